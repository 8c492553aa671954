#!/usr/bin/env python
"""Benchmark of the Monkey-Net frame-generation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without RANK in the environment: re-executes itself under `python -m torch.distributed.run --nnodes=1
     --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU over RCCL; the driver's own launch is left alone)

A "step" is one full training iteration of train.py:110-136 on one synthetic batch: KPDetector (source + driving
frame) -> generator -> discriminator on (generated, real) -> losses -> backward -> Adam (generator, kp detector),
then the discriminator update (mnk.engine.TrainStep: same losses and gradients as train.py's two passes, with one
discriminator forward serving both).  KPDetector / DenseMotionModule / generator and the discriminator run forward
and backward on the hand-written gfx950 kernels (libmonkeynet_hip.so), and so do the loss terms and the three Adam
steps (mnk.optim.MnkAdam: one launch per optimiser that also emits the packed weights of the next forward); what
is left on stock PyTorch-ROCm ops is autograd's own gradient accumulation and the batch means of the loss vectors
(SURVEY.md section 8f).  With several ranks the flat gradient buffer is all-reduced by the library's own RCCL
communicator and the SyncBN statistics travel through the peer-to-peer exchange inside the statistics kernels (csrc/p2p.hip;
RCCL when the mailboxes cannot be mapped: `syncbn_exchange` in the JSON line says which), inside the timed step.
Data: synthetic U[0,1) frame pairs (BASELINE.md section 2 protocol), random-init weights of the named configuration;
inputs are resident in HBM before the timed region.

One JSON line is printed by rank 0:  metric = train frames/sec (one frame = one generated driving frame = one
(source, driving) pair), whole-job aggregate over all ranks (weak scaling: fixed per-GPU batch), plus
  roofline      -- the conv implicit-GEMM kernels (every forward + data-gradient launch): algorithmic FLOPs (2*MAC of
                   the true, un-padded convolution) / kernel duration from HIP events stamped with each launch's own
                   begin and end on the launch stream (hipExtLaunchKernelGGL) in two profiled eager iterations after
                   the timed region, against the 157.3 TFLOP/s fp32-MFMA peak of MI355X_MICROARCH.md; `traffic` =
                   HBM bytes per launch from the committed PMC passes of the same build (profiles/, `traffic_source`);
                   `executed` / `executed_frac`: the multiply-adds the launches actually issue (the sub-pixel forms of the
                   up-sampled convolutions run 4/9 of the algorithmic ones) / time / peak;
  dropin        -- the reference's OWN loop on the drop-in modules (train.py:78-153's statement sequence: three
                   torch.optim.Adam, host batches through DataParallelWithCallback, per-iteration host copies of the losses).
                   Round 6: the wrappers serve it from three captured hipGraphs and step the stock optimisers with the library's
                   Adam kernel (mnk.dropin, `runner` = what served the calls); `with_mnk_adam`: the same loop with
                   mnk.optim.MnkAdam objects; `modules_as_they_are`: MNK_DROPIN_GRAPH=0 MNK_ADOPT_ADAM=0 (rounds 4-5's `dropin`:
                   two discriminator passes, eager launches); `eval_frame_loop`: reconstruction.py:45-62's per-frame loop at
                   batch 1 (frozen-weight hipGraph per wrapper) -- reported beside `value`, never as it;
  roofline_hbm  -- one record per HBM-bound kernel group (norm statistics / apply / backward, soft-argmax, movement embedding,
                   motion field, warps, 1x1 convolutions, Adam, layout, losses): algorithmic bytes per iteration / the HIP-event
                   kernel time of this run / 8 TB/s, `traffic` = counter bytes of the round's PMC passes when they were made
                   on these very sources;
  extra         -- the other single-GPU BASELINE configurations on the same clock: taichi_b32 (the stack the 0.5 target is
                   worded on), vox256_b8 (configs[3]'s per-GPU share), bair_b512_infer (configs[4]); bf16x3_gemm: the headline
                   workload with the opt-in fp32-accurate GEMM form on the bf16 matrix cores (tuning value gemm_bf16x3);
  hot_path_only_ms -- SURVEY section 8a alone (KPDetector + generator forward and backward with every weight gradient
                   materialised; no discriminator, losses or optimiser) as a hipGraph replay, next to the whole step;
  cpu_baseline  -- the CPU oracle (oracle/restate.py, a torch-CPU restatement of the reference; "port") timed on this
                   host's cores on a bounded sample of the same workload (rank 0, N == 1 only); its `recon_l1` = the
                   second half of BASELINE's metric: reconstruction L1 (reconstruction.py:74) of the eval forward on the
                   HIP path and on the oracle from the same weights and frames, and their difference (bound: 1e-4).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PMC_ROUND = "r06"               # profiles/<round>_pmc_traffic_<config>_b<batch>.json: the PMC passes `roofline.traffic` is read from


def kernel_source_stamp():
    """what a PMC traffic file must have been measured on to be quoted next to this run: the sources of the GEMM kernels"""
    import hashlib
    h = hashlib.sha256()
    for f in ("conv3x3.hip", "mnk_common.h", "pack_tile.h", "plan_table.h"):
        with open(os.path.join(ROOT, "monkey-net_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="moving-gif", help="moving-gif (BASELINE configs[1]) | taichi | shapes")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32, help="batch of the CPU sample (default: the workload's)")
    ap.add_argument("--cpu-steps", type=int, default=6, help="CPU sample: at most this many iterations / ~20 s")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--host-inputs", action="store_true",
                    help="additionally time K iterations whose inputs start in pinned host memory (PCIe-inclusive "
                         "rate, reported as `pcie_inclusive`; never `value`)")
    ap.add_argument("--dropin", type=int, default=1,
                    help="1 (default, one GPU): also time the reference's OWN loop on the drop-in modules -- train.py:78-153's "
                         "statement sequence (three torch.optim.Adam, host dict batches through DataParallelWithCallback("
                         "device_ids=[0]), two discriminator passes, per-iteration .cpu() of the losses) -- reported as "
                         "`dropin`, never as `value`; 0: skip")
    ap.add_argument("--taichi-leg", type=int, default=1,
                    help="1 (default, one GPU): also time the taichi @ 64x64 batch-32 hot path (the stack BASELINE.json's 0.5-of-"
                         "roofline target is worded on) -> `extra.taichi_b32`; 0: skip")
    ap.add_argument("--extra-legs", type=int, default=1,
                    help="1 (default): also measure BASELINE configs[3]'s per-GPU share (vox 256x256, batch 8: extra.vox256_b8) and "
                         "configs[4] (bair batch-512 inference: extra.bair_b512_infer) in the default run")
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: replay the iteration as one captured hipGraph, 0: eager launches, -1: graph on 1 GPU")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU check of the N-rank self-launch: every rank joins a gloo group, rank 0 prints one line")
    ap.add_argument("--print-launch", action="store_true", help="print the N-rank launch command and exit")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no rank environment: become the launcher -- one process per GPU under
    torch.distributed.run on this node (RCCL world size N), exactly the command the driver would have typed.  The
    reference's multi-GPU entry is likewise one command (run.py:28 `--device_ids`, train.py:104-105).  Returns only when
    this process already is a rank (RANK set) or N == 1."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    if not args.launcher_selftest:
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (args.gpus, have))
            sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    if args.print_launch:
        print(" ".join(cmd))
        sys.exit(0)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def launcher_selftest():
    """What the ranks of `--gpus N --launcher-selftest` run (CPU, gloo): proves the self-launch produced N ranks that can
    talk, and that rank 0 alone owns stdout."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    keep_stdout_for_json()
    if world > 1:
        dist.init_process_group(backend="gloo")
    t = torch.ones(1)
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        emit({"selftest": True, "n_gpus": world, "ranks_seen": int(t.item())})
    if world > 1:
        dist.destroy_process_group()


def build_models(cfg, device):
    from modules.generator import MotionTransferGenerator
    from modules.discriminator import Discriminator
    from modules.keypoint_detector import KPDetector
    mp = cfg["model_params"]
    torch.manual_seed(0)
    gen = MotionTransferGenerator(**mp["generator_params"], **mp["common_params"])
    disc = Discriminator(**mp["discriminator_params"], **mp["common_params"])
    kpd = KPDetector(**mp["kp_detector_params"], **mp["common_params"])
    return gen.to(device), disc.to(device), kpd.to(device)


def recon_l1_vs_cpu(cfg, size, device, batch=8):
    """"recon L1 vs CPU ref" (BASELINE.json's metric): the reconstruction criterion of reconstruction.py:74 -- mean
    |prediction - driving frame| of the eval-mode forward (key points of source and driving frame, generator) -- on the HIP
    path and on the CPU oracle (oracle/restate.py, fp32) from the same weights and frames.  The checker leg of the bench:
    nothing here is timed.  north_star's bound on |difference| is 1e-4."""
    from oracle import restate, cases
    from mnk import engine
    mp = cfg["model_params"]
    common = mp["common_params"]
    torch.manual_seed(1)
    gen, _, kpd = build_models(cfg, "cpu")
    for i, m in enumerate((gen, kpd)):       # non-trivial running statistics and affine parameters
        sd = m.state_dict()
        cases.perturb_state_dict(sd, 11 + i)
        m.load_state_dict(sd)
    sds = {"generator": {k: v.clone() for k, v in gen.state_dict().items()},
           "kp_detector": {k: v.clone() for k, v in kpd.state_dict().items()}}
    src, drv = cases.smooth_pair(batch, size, size)
    with torch.no_grad():
        kp_params = dict(mp["kp_detector_params"], **common)
        kp_s = restate.kp_detector_forward(sds["kp_detector"], kp_params, src, training=False)
        kp_d = restate.kp_detector_forward(sds["kp_detector"], kp_params, drv, training=False)
        ref = restate.generator_forward(sds["generator"], mp["generator_params"], common, src, kp_d, kp_s, training=False)
    out = engine.Reconstructor(kpd.to(device), gen.to(device))(src.to(device), drv.to(device))
    pred = out["video_prediction"].detach().cpu()
    l1_hip = float((pred.double() - drv.double()).abs().mean())
    l1_cpu = float((ref["video_prediction"].double() - drv.double()).abs().mean())
    return {"hip": l1_hip, "cpu_ref": l1_cpu, "abs_diff": abs(l1_hip - l1_cpu),
            "max_abs_frame_diff": float((pred - ref["video_prediction"]).abs().max()),
            "sample": "eval forward (kp detector x2 + generator), batch %d @ %dx%d, random-init weights with perturbed "
                      "BatchNorm statistics, smooth synthetic frames" % (batch, size, size)}


def reference_cpu_row(cfg_name, batch, size):
    """BASELINE.md section 2's row for this workload: the unmodified reference on the authoring container's host CPU"""
    try:
        rows = json.load(open(os.path.join(ROOT, "profiles", "r04_reference_cpu_timing.json")))["rows"]
        for r in rows:
            if r["config"] == cfg_name and r["batch"] == batch and r["size"] == size:
                return "%.2f frames/s (%.3f s/iteration, %d threads of %s)" % (r["frames_per_s"], r["s_per_step_mean"],
                                                                               r["threads"], r["cpu"])
    except Exception:
        pass
    return "no row for this workload (taichi batch 32: 12.8 pairs/s on 8 cores)"


def cpu_baseline(cfg, batch, size, steps, cfg_name=""):
    """The oracle restatement (torch CPU, fp32) doing the same training iteration: forward through
    restate.generator_full_forward / discriminator_full_forward, backward, 3x Adam."""
    from oracle import restate, cases
    from modules.generator import MotionTransferGenerator
    from modules.discriminator import Discriminator
    from modules.keypoint_detector import KPDetector
    mp, tp = cfg["model_params"], cfg["train_params"]
    torch.manual_seed(0)
    mods = {"generator": MotionTransferGenerator(**mp["generator_params"], **mp["common_params"]),
            "discriminator": Discriminator(**mp["discriminator_params"], **mp["common_params"]),
            "kp_detector": KPDetector(**mp["kp_detector_params"], **mp["common_params"])}
    # only the parameter containers are used here -- the arithmetic below is the oracle's
    sds = {k: {n: t.detach().clone().requires_grad_(t.is_floating_point() and "running" not in n)
               for n, t in m.state_dict().items()} for k, m in mods.items()}
    params = {k: [t for t in sd.values() if t.requires_grad] for k, sd in sds.items()}
    opts = {k: torch.optim.Adam(p, lr=tp["lr"], betas=(0.5, 0.999)) for k, p in params.items()}
    src, drv = cases.synthetic_pair(batch, size, size)
    # threads actually used: the host's cores, capped -- the small 64x64 ops of this net stop scaling (and start
    # thrashing in OpenMP barriers) long before a 100+ core host is filled
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)

    def one():
        losses, generated, kp_joined, ctx_g, ctx_k = restate.generator_full_forward(sds, cfg, src, drv)
        sum(v.mean() for v in losses).backward()
        opts["generator"].step(), opts["generator"].zero_grad(), opts["discriminator"].zero_grad()
        opts["kp_detector"].step(), opts["kp_detector"].zero_grad()
        with torch.no_grad():
            for ctx, key in ((ctx_g, "generator"), (ctx_k, "kp_detector")):
                for n, v in ctx.new_stats.items():
                    sds[key][n].copy_(v)
        dl = restate.discriminator_full_forward(sds, cfg, drv, {k: v.detach() for k, v in kp_joined.items()},
                                                {k: v.detach() for k, v in generated.items()})
        sum(v.mean() for v in dl).backward()
        opts["discriminator"].step(), opts["discriminator"].zero_grad()

    t0 = time.perf_counter()
    one()   # warm-up (also bounds the sample: if one iteration is already slow, it is the measurement)
    first = time.perf_counter() - t0
    done = 0
    t0 = time.perf_counter()
    while done < steps and (time.perf_counter() - t0) < 20.0 and first < 30.0:
        one()
        done += 1
    dt = (time.perf_counter() - t0) / done if done else first
    steps = max(done, 1)
    return {"value": batch / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d training iterations of the same config at batch %d (oracle/restate.py, torch CPU fp32, "
                      "%d threads), %.2f s/iteration.  A PORT: Conv2d on frames folded into the batch, not the reference's "
                      "own Conv3d((1,3,3)) modules (the reference tree cannot travel to the GPU box).  The UNMODIFIED reference "
                      "itself, timed in the authoring container by oracle/time_reference.py (BASELINE.md section 2, "
                      "profiles/r04_reference_cpu_timing.json): %s" % (steps, batch, cores, dt, reference_cpu_row(cfg_name, batch, size))}


def prof_collect(lib, fn, steps):
    """fn() `steps` times as real launches under the library's per-launch HIP-event recorder (mnk_prof_*): per kernel group
    {launches, ms, algorithmic work, executed work} per call of fn."""
    import ctypes
    lib.cdll.mnk_prof_reset()
    lib.cdll.mnk_prof_enable(1)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    lib.cdll.mnk_prof_enable(0)
    kernels = {}
    for k in range(lib.cdll.mnk_prof_num_kernels()):
        n, ms, work, issued = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        lib.cdll.mnk_prof_query(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work))
        lib.cdll.mnk_prof_query_executed(k, ctypes.byref(issued))
        if n.value:
            kernels[lib.cdll.mnk_prof_kernel_name(k).decode()] = {
                "launches_per_step": n.value / steps, "ms_per_step": ms.value / steps,
                "avg_us": ms.value / n.value * 1e3, "work_per_step": work.value / steps,
                "executed_work_per_step": issued.value / steps}
    lib.cdll.mnk_prof_reset()
    return kernels


def hot_path_record(hot_ms, hot_launch, hot_kernels, flops_total, batch):
    """The `hot_path_only` object of the JSON line: algorithmic rate (3 x the forward's FLOPs of the true convolutions / time)
    AND the executed one (the multiply-adds the GEMM launches of these very iterations issue -- the sub-pixel forms of the
    up-sampled convolutions run 4/9 of the algorithmic ones) side by side."""
    alg = 3 * flops_total * batch / (hot_ms * 1e-3) / 1e12
    rec = {"what": "KPDetector + generator forward and backward (all weight gradients), no discriminator / losses / optimiser",
           "launch": hot_launch, "ms": round(hot_ms, 3), "conv_tflops": round(alg, 2),
           "counts": "algorithmic FLOPs (3 x forward) over the replay's wall time; executed_*: what the matrix pipe issues",
           "frac_of_fp32_mfma_peak": round(alg / FP32_MFMA_PEAK_TFLOPS, 4)}
    if hot_kernels:
        conv = [hot_kernels[k] for k in ("conv3x3_igemm", "conv3x3_wgrad") if k in hot_kernels]
        issued = sum(g["executed_work_per_step"] for g in conv)
        counted = sum(g["work_per_step"] for g in conv)
        rec["executed_tflops"] = round(issued / (hot_ms * 1e-3) / 1e12, 2)
        rec["executed_frac"] = round(issued / (hot_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        rec["executed_gflop"] = round(issued / 1e9, 1)
        rec["algorithmic_gflop_counted_by_the_launches"] = round(counted / 1e9, 1)
        rec["gemm_kernel_ms"] = round(sum(g["ms_per_step"] for g in conv), 3)
    return rec


def hot_path_only(step, x, iters, device, lib=None):
    """the hot path alone -- KPDetector (source + driving frame) + generator forward, and their
    backward incl. every weight gradient, seeded with a fixed random dL/d(prediction); no discriminator, no losses, no
    optimiser.  Captured as one hipGraph like the full iteration and timed with HIP events over `iters` replays."""
    from mnk import engine
    gen, kpd = step.generator, step.kp_detector
    tp = step.tp
    g = torch.Generator().manual_seed(99)
    seed = torch.randn(x["video"].shape, generator=g).to(device)
    opts = [o for o in (step.opt_g, step.opt_k) if hasattr(o, "materialize_grads")]

    def one():
        kp_joined = kpd(torch.cat([x["source"], x["video"]], dim=2))
        out = gen(x["source"], **engine.split_kp(kp_joined, tp["detach_kp_generator"]))
        torch.autograd.backward([out["video_prediction"]], [seed])
        for o in opts:
            o.materialize_grads()
        step.opt_g.zero_grad(), step.opt_k.zero_grad()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            one()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize(device)
    launch = "hipGraph replay"
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            one()
        run = graph.replay
    except Exception as e:      # pragma: no cover - capture is an optimisation
        sys.stderr.write("hot-path-only capture failed (%s); timing eager launches\n" % type(e).__name__)
        run, launch = one, "eager"
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / iters
    if lib is None:
        return ms, launch
    return ms, launch, prof_collect(lib, one, 2)      # the same iteration as real launches: what its GEMMs issue


def taichi_b32_leg(lib, device, steps):
    """BASELINE.json's target is worded on "the 64x64 generator conv stack at batch 32" with the taichi hyper-parameters
    (174.9 GFLOP forward); the headline workload of this file is moving-gif (configs[1]).  This leg puts
    that stack on the same clock: the hot path alone as a hipGraph replay, algorithmic and executed fraction of the fp32 MFMA
    peak side by side, and the whole training iteration of the configuration next to it."""
    from mnk import configs, engine, workload
    cfg = configs.get("taichi")
    gen, disc, kpd = build_models(cfg, device)
    src, drv = workload.synthetic_pair(32, 64, 64, seed=4321)
    x = {"source": src.to(device), "video": drv.to(device)}
    eager = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=False)
    for o in (eager.opt_g, eager.opt_d, eager.opt_k):
        if hasattr(o, "reducer"):
            o.reducer.bg_macs = 0.0          # (profiled iterations: no weight-gradient GEMMs running under other kernels)
    for _ in range(2):
        eager.step(x)
    torch.cuda.synchronize(device)
    hot_ms, hot_launch, hot_k = hot_path_only(eager, x, max(steps, 10), device, lib)
    flops = workload.conv_flops_hot_path(cfg, 64, 64)
    out = {"workload": "taichi model params @ 64x64, batch 32, one GPU",
           "hot_path_conv_gflop_fwd_per_pair": round(flops["total"] / 1e9, 3),
           "hot_path_only": hot_path_record(hot_ms, hot_launch, hot_k, flops["total"], 32)}
    conv = hot_k.get("conv3x3_igemm")
    if conv:
        out["roofline"] = {"kernel": "conv3x3_igemm*: forward + data-gradient launches of the hot path", "bound": "mfma",
                           "achieved": round(conv["work_per_step"] / (conv["ms_per_step"] * 1e-3) / 1e12, 2),
                           "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(conv["work_per_step"] / (conv["ms_per_step"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                           "executed_frac": round(conv["executed_work_per_step"] / (conv["ms_per_step"] * 1e-3) / 1e12
                                                  / FP32_MFMA_PEAK_TFLOPS, 4),
                           "launches": conv["launches_per_step"], "avg_launch_us": round(conv["avg_us"], 2)}
    try:
        gen2, disc2, kpd2 = build_models(cfg, device)
        step = engine.TrainStep(gen2, disc2, kpd2, cfg["train_params"], use_graph=True)
        for _ in range(3):
            step.step(x)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(max(steps, 10)):
            step.step(x)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / max(steps, 10)
        out["full_iteration"] = {"ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(32 / dt, 1), "launch": "hipGraph replay"}
    except Exception as e:    # never lose the leg to the extra number
        out["full_iteration"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s (about 6.3 TB/s is reachable by a streaming kernel)

# kernel-name prefixes of each HIP-event group (csrc/runtime.hip::kNames): how the rows of a PMC traffic file map onto the groups
HBM_GROUPS = {
    "bn_stats": ("colsum2_", "bn_final_finalize", "bn_finalize", "bn_small_fwd"),
    "bn_act_apply": ("bn_act_fwd", "inorm_act_fwd"),
    "bn_act_bwd": ("bn_act_bwd", "bn_small_bwd", "inorm_act_bwd"),
    "softmax_kp": ("softmax_kp_", "kp_clip_var", "heatmap_argmax", "kp_pixel_index"),
    "movement_embedding": ("movement_embedding_", "gaussian_sums"),
    "motion_field": ("motion_field_",),
    "deform": ("warp_levels_", "deform_"),
    "conv1x1": ("gconv1x1_", "conv1x1_"),
    "adam_pack": ("adam_multi", "adam_tick"),
    "layout": ("ncdhw_to_nhwc", "nhwc_to_ncdhw", "copy_channels", "concat2_", "sumpool2x2", "resize_", "pair_l1_"),
    "losses": ("l1_mean_", "gan_terms_", "vec_means_"),
}


def all_source_stamp():
    """hash of EVERY kernel source: what a PMC file must have been measured on for the traffic of the non-GEMM kernels to be quoted"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "monkey-net_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "monkey-net_amd", "csrc", "*.h"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def roofline_hbm(kernels, pmc=None, adam_params=0):
    """One record per HBM-bound kernel group of the iteration (north_star: "evidenced by rocprof HBM GB/s"): ALGORITHMIC bytes
    per iteration -- what each launch declares to the library's recorder (SURVEY.md section 8d's per-unit figures: 4 C B/pixel
    read + written by an apply pass, one read of the heat-map for the soft-argmax, 28 B per parameter for Adam ...; DESIGN.md
    section 4) -- over the HIP-event kernel time of THIS run, against the 8 TB/s HBM3E peak; `traffic`: counter bytes per
    iteration from the round's PMC file when it was measured on these very sources, else null."""
    out = []
    for name, prefixes in HBM_GROUPS.items():
        g = kernels.get(name)
        if not g or not g["ms_per_step"]:
            continue
        work = g["work_per_step"]
        if name == "adam_pack" and adam_params:
            work = 28.0 * adam_params          # read p, g, m, v; write p, m, v (the packs it also emits are an artefact of the GEMM layouts)
        gbps = work / (g["ms_per_step"] * 1e-3) / 1e9
        rec = {"kernel": name, "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
               "frac": round(gbps / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_step": int(work),
               "launches_per_step": g["launches_per_step"], "ms_per_step": round(g["ms_per_step"], 4),
               "avg_launch_us": round(g["avg_us"], 2), "traffic": None}
        if pmc:
            n = t = 0.0
            for k, v in pmc.items():
                if isinstance(v, dict) and k.startswith(prefixes):
                    n += v["launches"]
                    t += (2 * v["fetch_bytes_per_launch_raw"] + v["write_bytes_per_launch"]) * v["launches"]
            if n and pmc.get("_iterations"):
                rec["traffic"] = int(t / pmc["_iterations"])
                rec["traffic_frac_of_peak"] = round(t / pmc["_iterations"] / (g["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        out.append(rec)
    return out


def vox256_b8_leg(lib, device, steps):
    """BASELINE configs[3] (vox 256x256, batch 64 over 8 GPUs) on ONE GPU's share: batch 8 at 256x256, the whole training
    iteration as a hipGraph replay + the forward / data-gradient GEMMs' roofline from two eager iterations."""
    from mnk import configs, engine, workload
    cfg = configs.get("vox")
    gen, disc, kpd = build_models(cfg, device)
    src, drv = workload.synthetic_pair(8, 256, 256, seed=777)
    x = {"source": src.to(device), "video": drv.to(device)}
    out = {"workload": "vox model params @ 256x256, batch 8 (one GPU's share of BASELINE configs[3]), full train.py:110-136 iteration"}
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=True)
    for _ in range(3):
        step.step(x)
    torch.cuda.synchronize(device)
    n = max(5, min(steps, 10))
    t0 = time.perf_counter()
    for _ in range(n):
        step.step(x)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / n
    out["full_iteration"] = {"ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(8 / dt, 1), "launch": "hipGraph replay"}
    eager = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=False)
    for o in (eager.opt_g, eager.opt_d, eager.opt_k):
        if hasattr(o, "reducer"):
            o.reducer.bg_macs = 0.0          # no weight-gradient GEMMs running under the kernels that are being timed
    eager.step(x)
    k = prof_collect(lib, lambda: eager.step(x), 2)
    conv = k.get("conv3x3_igemm")
    if conv:
        ach = conv["work_per_step"] / (conv["ms_per_step"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": "conv3x3_igemm*: forward + data-gradient launches", "bound": "mfma", "achieved": round(ach, 2),
                           "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                           "executed_frac": round(conv["executed_work_per_step"] / (conv["ms_per_step"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                           "launches": conv["launches_per_step"], "avg_launch_us": round(conv["avg_us"], 2)}
    out["roofline_hbm"] = roofline_hbm(k, adam_params=sum(p.numel() for m in (gen, disc, kpd) for p in m.parameters()))
    flops = workload.conv_flops_hot_path(cfg, 256, 256)
    out["hot_path_conv_gflop_fwd_per_pair"] = round(flops["total"] / 1e9, 3)
    return out


def bair_b512_infer_leg(device, iters=8):
    """BASELINE configs[4]: bair 64x64 reconstruction-mode inference at batch 512 (reconstruction.py:12-25,57-62 batched by
    mnk.engine.Reconstructor): eager launches and the hipGraph-captured forward.  35 ms of long launches per batch: a replay buys
    nothing here (the eager loop already keeps the GPU busy), so both are reported and `value` is the better one."""
    from mnk import configs, engine, workload
    cfg = configs.get("bair")
    gen, _, kpd = build_models(cfg, device)
    g = torch.Generator().manual_seed(5)
    src = torch.rand(512, 3, 1, 64, 64, generator=g).to(device)
    drv = torch.rand(512, 3, 1, 64, 64, generator=g).to(device)
    res = {"workload": "bair eval forward (kp detector on source + driving, generator), batch 512 @ 64x64 (BASELINE configs[4])"}
    for mode in ("eager", "graph"):
        r = engine.Reconstructor(kpd, gen, use_graph=(mode == "graph"))
        for _ in range(2):
            out = r(src, drv)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(iters):
            out = r(src, drv)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / iters
        res[mode] = {"ms_per_batch": round(dt * 1e3, 3), "frames_per_s": round(512 / dt, 1),
                     "finite": bool(torch.isfinite(out["video_prediction"]).all())}
    best = max(("eager", "graph"), key=lambda m: res[m]["frames_per_s"])
    res["value"], res["unit"], res["launch"] = res[best]["frames_per_s"], "frames/s", best
    flops = workload.conv_flops_hot_path(cfg, 64, 64)["total"]
    res["conv_tflops"] = round(flops * 512 / (res[best]["ms_per_batch"] * 1e-3) / 1e12, 1)
    res["frac_of_fp32_mfma_peak"] = round(res["conv_tflops"] / FP32_MFMA_PEAK_TFLOPS, 4)
    return res


def bf16x3_leg(lib, cfg, x, device, steps):
    """The SAME workload as the headline with the library's tuning values `gemm_bf16x3` = `wgrad_bf16x3` = 1: the forward /
    data-gradient GEMMs of the 32x32-tile implicit-GEMM kernels and the tap-major weight-gradient GEMMs run their fp32 products on
    the bf16 matrix cores -- both fp32 operands split exactly into
    three bf16 terms by the loaders, six v_mfma_f32_32x32x16_bf16 per K step, fp32 accumulation (csrc/mnk_common.h).  An
    fp32-accurate product (same error against fp64 as the fp32 MFMA chain: tests/test_kernels_conv*.py, tools/microbench/
    bf16x3_gemm.hip), reported BESIDE the headline, which stays on v_mfma_f32_32x32x2_f32: opt-in until the weight-gradient and
    16x16-tile kernels have their forms too."""
    from mnk import engine
    lib.call("mnk_set_tuning", b"gemm_bf16x3", 1)
    lib.call("mnk_set_tuning", b"wgrad_bf16x3", 1)
    try:
        gen, disc, kpd = build_models(cfg, device)
        step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=True)
        for _ in range(3):
            step.step(x)
        torch.cuda.synchronize(device)
        n = max(steps, 10)
        t0 = time.perf_counter()
        for _ in range(n):
            out = step.step(x)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / n
        b = int(x["source"].shape[0])
        rec = {"ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(b / dt, 1), "launch": "hipGraph replay",
               "finite": bool(all(float(v) == float(v) for v in out[0])),
               "what": "MNK_TUNING=gemm_bf16x3=1,wgrad_bf16x3=1: the forward / data-gradient GEMMs of the 32x32-tile kernels and the "
                       "tap-major weight-gradient GEMMs as six bf16 MFMAs per K step on the exact three-way bf16 split of both fp32 "
                       "operands (fp32 accumulation; fp32-accurate: same error against fp64 as the fp32 MFMA chain).  Not the "
                       "headline: opt-in this round"}
        eager = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=False)
        for o in (eager.opt_g, eager.opt_d, eager.opt_k):
            if hasattr(o, "reducer"):
                o.reducer.bg_macs = 0.0      # as in the headline's profiled iterations: no weight-gradient GEMMs under other kernels
        eager.step(x)
        k = prof_collect(lib, lambda: eager.step(x), 2)
        conv = k.get("conv3x3_igemm")
        if conv:
            rec["conv3x3_igemm"] = {"ms_per_step": round(conv["ms_per_step"], 3), "launches_per_step": conv["launches_per_step"],
                                    "algorithmic_tflops": round(conv["work_per_step"] / (conv["ms_per_step"] * 1e-3) / 1e12, 2),
                                    "executed_tflops": round(conv["executed_work_per_step"] / (conv["ms_per_step"] * 1e-3) / 1e12, 2),
                                    "note": "fp32-equivalent FLOPs; the group still holds the 16x16-tile fp32 launches (45-channel "
                                            "refinement stack, narrow heads)"}
        return rec
    finally:
        lib.call("mnk_set_tuning", b"gemm_bf16x3", 0)
        lib.call("mnk_set_tuning", b"wgrad_bf16x3", 0)


def dropin_loop(cfg, x, device, steps, warmup, mnk_adam=False):
    """What a user of the reference gets by putting monkey-net_amd/ in front of the reference root and running the reference's
    unmodified train.py: its loop (train.py:78-153), statement for statement as tests/test_dropin_replay.py restates it --
    three torch.optim.Adam(betas=(0.5, 0.999)), the two full models (train.py:24-75 = mnk.engine.GeneratorFullModel /
    DiscriminatorFullModel) behind DataParallelWithCallback(device_ids=[0]), HOST dict batches as a DataLoader hands them
    over (pinned; the wrapper moves them), generator pass -> backward -> steps, discriminator pass (a second discriminator
    forward) -> backward -> step, and the per-iteration host copies of the loss values (train.py:125,138).  No hipGraph, no
    MnkAdam, no shared discriminator forward: every launch is issued by the Python loop.
    mnk_adam: the same loop with train.py:81-83's three `torch.optim.Adam(...)` replaced by `mnk.optim.MnkAdam(...)` -- the one
    edit of the reference's loop INTEGRATION.md section 1.5 offers (both variants: tests/test_dropin_replay.py)."""
    from mnk.engine import GeneratorFullModel, DiscriminatorFullModel
    from mnk.optim import MnkAdam
    from sync_batchnorm import DataParallelWithCallback
    tp = cfg["train_params"]
    generator, discriminator, kp_detector = build_models(cfg, device)
    Adam = MnkAdam if mnk_adam else torch.optim.Adam
    opt_g = Adam(generator.parameters(), lr=tp["lr"], betas=(0.5, 0.999))
    opt_d = Adam(discriminator.parameters(), lr=tp["lr"], betas=(0.5, 0.999))
    opt_k = Adam(kp_detector.parameters(), lr=tp["lr"], betas=(0.5, 0.999))
    gpar = DataParallelWithCallback(GeneratorFullModel(kp_detector, generator, discriminator, tp), device_ids=[0])
    dpar = DataParallelWithCallback(DiscriminatorFullModel(kp_detector, generator, discriminator, tp), device_ids=[0])
    host = {k: v.cpu().pin_memory() for k, v in x.items()}

    def iteration():
        xb = dict(host)
        out = gpar(xb)
        loss_values = [val.mean() for val in out[:-2]]
        generated, kp_joined = out[-2], out[-1]
        sum(loss_values).backward(retain_graph=not tp["detach_kp_discriminator"])
        opt_g.step(), opt_g.zero_grad(), opt_d.zero_grad()
        if tp["detach_kp_discriminator"]:
            opt_k.step(), opt_k.zero_grad()
        g_host = [val.detach().cpu().numpy() for val in loss_values]
        loss_values = [val.mean() for val in dpar(xb, kp_joined, generated)]
        sum(loss_values).backward()
        opt_d.step(), opt_d.zero_grad()
        if not tp["detach_kp_discriminator"]:
            opt_k.step(), opt_k.zero_grad()
        return g_host + [val.detach().cpu().numpy() for val in loss_values]

    for _ in range(warmup):
        iteration()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        last = iteration()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    b = int(x["source"].shape[0])
    from mnk import dropin, optim as moptim
    runner = dropin.runner_for(gpar.module)
    served = None if runner is None else dict(runner.stats)
    how = "the wrapped modules called as they are (two discriminator passes), eager launches"
    if served and served["graph_calls"]:
        how = ("forward / loss.backward() / discriminator pass served from three captured hipGraphs behind whole-model autograd "
               "Functions (mnk.dropin.TrainPairRunner; one discriminator forward per iteration)")
        if not mnk_adam:
            how += "; the stock optimisers stepped by mnk_adam_multi on their own state tensors (mnk.optim.AdoptedAdam): %s" % (
                [getattr(moptim.adopted(o), "steps_taken", 0) for o in (opt_g, opt_d, opt_k)],)
    return {"value": round(b / dt, 2), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3),
            "finite": bool(all(float(v) == float(v) for v in last)),
            "runner": served,
            "what": "the reference's own loop on the drop-in modules: train.py:78-153 statement sequence (3x %s, "
                    "host batches through DataParallelWithCallback(device_ids=[0]), host copies of the losses every iteration); %s "
                    "-- tests/test_dropin_replay.py is its parity test" % ("mnk.optim.MnkAdam" if mnk_adam else "torch.optim.Adam", how)}


def frame_loop(cfg, size, device, frames=40):
    """The reference's per-frame evaluation loop on the drop-in modules (reconstruction.py:45-62: for every frame of a video,
    kp_detector(frame) and generator(source, kp_driving, kp_source) at batch 1 under no_grad, behind DataParallelWithCallback):
    milliseconds per frame.  Round 6: each wrapper replays a frozen-weight hipGraph per input signature (mnk.dropin.EvalRunner;
    MNK_EVAL_GRAPH=0: eager launches).  (mnk.engine.Reconstructor is the batched form of the same work.)"""
    from sync_batchnorm import DataParallelWithCallback
    gen, _, kpd = build_models(cfg, device)
    generator, kp_detector = DataParallelWithCallback(gen), DataParallelWithCallback(kpd)
    generator.eval(), kp_detector.eval()
    video = torch.rand(1, 3, frames, size, size)

    def loop():
        with torch.no_grad():
            kp_source = kp_detector(video[:, :, :1])
            for i in range(frames):
                kp_driving = kp_detector(video[:, :, i:i + 1])
                out = generator(source_image=video[:, :, :1], kp_driving=kp_driving, kp_source=kp_source)
        return out["video_prediction"]

    loop()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    last = loop()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / frames
    from mnk import dropin
    runners = [dropin.eval_runner_for_wrapper(w) for w in (kp_detector, generator)]
    served = all(r is not None and r.stats["replays"] > 0 for r in runners)
    return {"ms_per_frame": round(dt * 1e3, 3), "frames_per_s": round(1.0 / dt, 1), "finite": bool(torch.isfinite(last).all()),
            "launch": "frozen-weight hipGraph replay per wrapper call (mnk.dropin.EvalRunner)" if served else "eager",
            "what": "reconstruction.py:45-62's loop on the drop-in modules: kp_detector + generator per frame at batch 1, "
                    "no_grad, host frames through DataParallelWithCallback"}


_JSON_FD = [None]


def keep_stdout_for_json():
    """RCCL prints a version banner on stdout when the communicator comes up; the contract is ONE JSON line there.
    File descriptor 1 points at stderr until emit() writes the line to the real stdout."""
    sys.stdout.flush()
    _JSON_FD[0] = os.dup(1)
    os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    if _JSON_FD[0] is None:
        os.write(1, line)
    else:
        os.write(_JSON_FD[0], line)


def graph_phase(args, rank, fallback, run, device):
    """Multi-rank only: capture + time the hipGraph iteration under a deadline.  Returns the elapsed seconds of the K
    timed replays, or None when any rank failed to capture.  If the phase does not finish within MNK_GRAPH_DEADLINE_S
    (a collective stuck inside a replay cannot be recovered in-process), rank 0 prints the already measured eager line
    and every rank leaves -- the bench never hangs on the optimisation."""
    import threading

    def expire():
        sys.stderr.write("rank %d: hipGraph phase exceeded its deadline; reporting the eager measurement\n" % rank)
        if rank == 0:
            if fallback is not None:
                fallback["capture_failed"] = True
                fallback["config"]["launch"] = "eager (the hipGraph phase exceeded MNK_GRAPH_DEADLINE_S)"
            emit(fallback)
        os._exit(0)

    guard = threading.Timer(float(os.environ.get("MNK_GRAPH_DEADLINE_S", "150")), expire)
    guard.daemon = True
    guard.start()
    dt, ok = None, 1
    try:
        dt = run()
    except Exception as e:      # capture is an optimisation, never a requirement
        sys.stderr.write("rank %d: hipGraph capture failed (%s: %s); keeping the eager measurement\n"
                         % (rank, type(e).__name__, e))
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # the ranks must agree on which measurement is reported
    all_ok = int(flag.item()) == 1
    guard.cancel()
    return dt if all_ok else None


def main():
    args = parse()
    # a benchmark has no checkpoint or visualiser that makes a rank late: a peer that is given up on should cost seconds
    os.environ.setdefault("MNK_P2P_TIMEOUT_MS", "15000")
    self_launch(args)
    if args.launcher_selftest:
        return launcher_selftest()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("MNK_DIST_FORCE", "") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        keep_stdout_for_json()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)

    from mnk import configs, engine, _lib
    from mnk import workload
    cfg = configs.get(args.config)
    lib = _lib.lib()
    assert lib.is_device_build, "bench.py must run on the real HIP library"
    gen, disc, kpd = build_models(cfg, device)
    # default: the whole iteration (with its RCCL collectives when there are several ranks) replays as one hipGraph
    use_graph = (os.environ.get("MNK_DIST_GRAPH", "1") == "1" or (world == 1 and not force_dist)) \
        if args.graph < 0 else bool(args.graph)
    src, drv = workload.synthetic_pair(args.batch, args.size, args.size, seed=1234 + rank)
    x = {"source": src.to(device), "video": drv.to(device)}
    dist_mode = world > 1 or force_dist

    def sync():
        torch.cuda.synchronize(device)
        if dist_mode:
            dist.barrier()
            torch.cuda.synchronize(device)

    def timed(st):
        """W un-timed + exactly K timed iterations, barrier + synchronize on both sides, MAX over the ranks."""
        for _ in range(args.warmup):
            st.step(x)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.step(x)
        sync()
        dt = time.perf_counter() - t0
        if dist_mode:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    capture_failed = False       # a requested hipGraph capture that did not happen: loud in the JSON line, not only on stderr
    p2p_fell_back = 0
    if not dist_mode:
        step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=use_graph)
        if use_graph:
            try:
                step.step(x)
                torch.cuda.synchronize(device)
            except Exception as e:   # capture is an optimisation, never a requirement
                sys.stderr.write("hipGraph capture failed (%s: %s); running eager launches\n" % (type(e).__name__, e))
                use_graph = False
                capture_failed = True
                torch.cuda.synchronize(device)
                gen, disc, kpd = build_models(cfg, device)
                step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=False)
        eager = step if not use_graph else engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=False)
        elapsed = timed(step)
        launch = "hipGraph replay" if use_graph else "eager"
    else:
        # several ranks: the eager iteration is measured FIRST (a complete, valid K-step measurement), then the iteration
        # with its RCCL collectives is captured as a hipGraph and measured again under a deadline (below) -- a capture
        # problem on some RCCL / driver combination costs the graph number, never the bench line
        eager = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=False)
        elapsed = timed(eager)
        launch = "eager"
        # The peer-to-peer SyncBN exchange passed its start-up self-test, but a first run on N GPUs is a first run: if an
        # exchange of the measured iterations gave a peer up (the sums are NaN then, mnk.dist.p2p_error() says which rank), every
        # rank leaves the peer-to-peer path, the models are rebuilt and the measurement is repeated on the collective path
        # (one RCCL all-reduce per norm layer and direction) -- a valid line instead of an invalid one, and the JSON says so.
        from mnk import dist as mdist
        code = torch.tensor([mdist.p2p_error()], dtype=torch.int32, device=device)
        dist.all_reduce(code, op=dist.ReduceOp.MAX)
        p2p_fell_back = int(code.item())
        if p2p_fell_back:
            sys.stderr.write("rank %d: a peer-to-peer SyncBN exchange gave rank %d up; repeating the measurement on the "
                             "collective path\n" % (rank, p2p_fell_back - 1))
            mdist.disable_p2p()
            gen, disc, kpd = build_models(cfg, device)
            eager = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=False)
            elapsed = timed(eager)
    ms_per_step = elapsed / args.steps * 1e3
    global_batch = args.batch * world
    value = global_batch * args.steps / elapsed

    pcie = None
    if args.host_inputs and not dist_mode:
        # the reference's loop hands over DataLoader (host) batches; here: pinned host tensors, one asynchronous
        # host-to-device copy per iteration on the launch stream, then the same step
        host = {k: v.cpu().pin_memory() for k, v in x.items()}
        for _ in range(args.warmup):
            step.step({k: v.to(device, non_blocking=True) for k, v in host.items()})
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step.step({k: v.to(device, non_blocking=True) for k, v in host.items()})
        sync()
        dt = time.perf_counter() - t0
        pcie = {"value": round(global_batch * args.steps / dt, 2), "unit": "frames/s",
                "ms_per_step": round(dt / args.steps * 1e3, 3),
                "bytes_per_step": int(sum(v.numel() * 4 for v in host.values()))}

    # ---- per-kernel HIP-event timing of one extra (un-timed) step: roofline of the dominant kernel --------------
    roofline = None
    kernels = {}
    if not args.no_profile:      # every rank runs the profiled steps (they contain the SyncBN / gradient collectives)
        for o in (getattr(eager, "opt_g", None), getattr(eager, "opt_d", None), getattr(eager, "opt_k", None)):
            if hasattr(o, "reducer"):
                o.reducer.bg_macs = 0.0      # as in a captured iteration: no weight-gradient GEMMs running under other kernels
        # event timing needs real launches (a graph replay bypasses the recorder)
        kernels = prof_collect(lib, lambda: eager.step(x), 2)
        conv = kernels.get("conv3x3_igemm")
        traffic, traffic_src = None, None
        pmc_file = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s_b%d.json" % (PMC_ROUND, args.config, args.batch))
        if os.path.exists(pmc_file) and args.size == 64:
            # HBM-side bytes per launch of the same kernel on the same workload, from separate rocprofv3 --pmc passes
            # (FETCH_SIZE, WRITE_SIZE; tools/gpu_evidence.sh + tools/pmc_summarize.py), gfx950 correction: 2 x FETCH_SIZE.
            # A number from a profile run, not from this run: it is quoted ONLY when that run was made on the GEMM kernel
            # sources of this run (`_kernel_source_stamp`); otherwise traffic stays null and `traffic_source` says why.
            pm = json.load(open(pmc_file))
            if pm.get("_kernel_source_stamp") != kernel_source_stamp():
                traffic_src = "%s was measured on other kernel sources (%s, stamp %s; this run: %s): not quoted" % (
                    os.path.basename(pmc_file), pm.get("_measured_on", "commit not recorded"),
                    pm.get("_kernel_source_stamp"), kernel_source_stamp())
                pm = {}
            n = f = w = 0.0
            for k, v in pm.items():
                if isinstance(v, dict) and "conv3x3_igemm" in k:
                    n += v["launches"]
                    f += v["fetch_bytes_per_launch_raw"] * v["launches"]
                    w += v["write_bytes_per_launch"] * v["launches"]
            if n:
                traffic = round((2 * f + w) / n)
                traffic_src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of %s (%s, kernel sources %s = this run's)" % (
                    os.path.basename(pmc_file), pm.get("_measured_on", "commit not recorded"), pm["_kernel_source_stamp"])
        if conv:
            achieved = conv["work_per_step"] / (conv["ms_per_step"] * 1e-3) / 1e12
            executed = conv["executed_work_per_step"] / (conv["ms_per_step"] * 1e-3) / 1e12
            roofline = {"kernel": "conv3x3_igemm_kernel / conv3x3_igemm16_kernel: every forward + data-gradient launch "
                                  "of the iteration (3x3 hot path and the discriminator's 4x4 convolutions)",
                        "bound": "mfma",
                        "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                        # what the matrix pipe does: the multiply-adds actually ISSUED (the sub-pixel forms of the up-sampled
                        # convolutions run 4/9 of the algorithmic ones `achieved` / `frac` credit) / time / peak
                        "executed": round(executed, 2), "executed_frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": traffic,
                        "traffic_source": traffic_src,
                        "launches_per_step": conv["launches_per_step"], "avg_launch_us": round(conv["avg_us"], 2),
                        "flop_per_launch": conv["work_per_step"] / conv["launches_per_step"]}
    # every convolution launch of the iteration, not only the best group: forward + data gradient (conv3x3_igemm), weight
    # gradient GEMMs (conv3x3_wgrad) and EVERY split reduction / weight re-layout launch that exists because of how those
    # GEMMs are tiled (conv3x3_reduce_pack: split-K reductions, the weight-gradient partial reduction, packs)
    roofline_all = None
    if kernels.get("conv3x3_igemm") and kernels.get("conv3x3_wgrad"):
        groups = [kernels[k] for k in ("conv3x3_igemm", "conv3x3_wgrad", "conv3x3_reduce_pack") if k in kernels]
        work = kernels["conv3x3_igemm"]["work_per_step"] + kernels["conv3x3_wgrad"]["work_per_step"]
        issued = kernels["conv3x3_igemm"]["executed_work_per_step"] + kernels["conv3x3_wgrad"]["executed_work_per_step"]
        ms = sum(g["ms_per_step"] for g in groups)
        roofline_all = {"what": "forward + data-gradient + weight-gradient GEMMs and every split reduction / pack launch",
                        "bound": "mfma", "achieved": round(work / (ms * 1e-3) / 1e12, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(work / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                        "executed": round(issued / (ms * 1e-3) / 1e12, 2),
                        "executed_frac": round(issued / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                        "gflop_per_step": round(work / 1e9, 1), "executed_gflop_per_step": round(issued / 1e9, 1),
                        "ms_per_step": round(ms, 3),
                        "launches_per_step": sum(g["launches_per_step"] for g in groups)}
    hbm = None
    if kernels:
        pm_all = None
        if os.path.exists(pmc_file) and args.size == 64:
            pm_all = json.load(open(pmc_file))
            if pm_all.get("_all_source_stamp") != all_source_stamp():
                pm_all = None               # measured on other kernel sources: the counter bytes are not quoted
        hbm = roofline_hbm(kernels, pm_all, adam_params=sum(p.numel() for m in (gen, disc, kpd) for p in m.parameters()))
    dropin = None
    if args.dropin and not dist_mode and rank == 0:
        try:
            dropin = dropin_loop(cfg, x, device, max(5, min(args.steps, 20)), 3)
            m = dropin_loop(cfg, x, device, max(5, min(args.steps, 20)), 3, mnk_adam=True)
            dropin["with_mnk_adam"] = {"value": m["value"], "unit": m["unit"], "ms_per_step": m["ms_per_step"], "finite": m["finite"],
                                       "runner": m["runner"],
                                       "what": "the same loop with train.py:81-83's three torch.optim.Adam replaced by "
                                               "mnk.optim.MnkAdam (INTEGRATION.md section 1.5)"}
            # the same loop with the runner and the optimiser adoption switched off: what rounds 4-5 reported as `dropin`
            os.environ["MNK_DROPIN_GRAPH"], os.environ["MNK_ADOPT_ADAM"] = "0", "0"
            try:
                e = dropin_loop(cfg, x, device, max(5, min(args.steps, 20)), 3)
                dropin["modules_as_they_are"] = {"value": e["value"], "unit": e["unit"], "ms_per_step": e["ms_per_step"],
                                                 "finite": e["finite"], "what": "MNK_DROPIN_GRAPH=0 MNK_ADOPT_ADAM=0: " + e["what"]}
            finally:
                os.environ.pop("MNK_DROPIN_GRAPH", None), os.environ.pop("MNK_ADOPT_ADAM", None)
        except Exception as e:   # never lose the bench line to the extra measurement
            dropin = {"error": "%s: %s" % (type(e).__name__, e)}
        if isinstance(dropin, dict) and "error" not in dropin:
            try:
                dropin["eval_frame_loop"] = frame_loop(cfg, args.size, device)
            except Exception as e:
                dropin["eval_frame_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}
    hot_ms, hot_launch, hot_kernels = None, None, None
    extra = {}
    if not args.no_profile and not dist_mode:
        try:
            hot_ms, hot_launch, hot_kernels = hot_path_only(eager, x, max(args.steps, 10), device, lib)
        except Exception as e:   # never lose the bench line to the extra measurement
            sys.stderr.write("hot-path-only measurement failed: %s: %s\n" % (type(e).__name__, e))
        if args.taichi_leg and rank == 0 and not (args.config == "taichi" and args.batch == 32 and args.size == 64):
            try:
                extra["taichi_b32"] = taichi_b32_leg(lib, device, args.steps)
            except Exception as e:
                extra["taichi_b32"] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
        default_workload = args.config == "moving-gif" and args.batch == 32 and args.size == 64
        if args.extra_legs and rank == 0 and default_workload:
            for name, leg in (("vox256_b8", lambda: vox256_b8_leg(lib, device, args.steps)),
                              ("bair_b512_infer", lambda: bair_b512_infer_leg(device)),
                              ("bf16x3_gemm", lambda: bf16x3_leg(lib, cfg, x, device, args.steps))):
                try:
                    extra[name] = leg()
                except Exception as e:   # never lose the bench line to an extra leg
                    extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                torch.cuda.empty_cache()
    if world > 1 or force_dist:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, args.cpu_batch, args.size, args.cpu_steps, args.config)
        try:        # the second half of BASELINE's metric; never lose the bench line to it
            cpu["recon_l1"] = recon_l1_vs_cpu(cfg, args.size, device)
        except Exception as e:
            cpu["recon_l1"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        flops = workload.conv_flops_hot_path(cfg, args.size, args.size)
        out = {
            "metric": "train frames/sec (Bx3x%dx%d, %d kp)" % (args.size, args.size,
                                                               cfg["model_params"]["common_params"]["num_kp"]),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s model params @ %dx%d, batch %d/GPU, full train.py:110-136 iteration "
                                   "(G step + D step, 3x Adam)" % (args.config, args.size, args.size, args.batch),
                       "global_batch": global_batch, "parallelism": "dp%d" % world,
                       "launch": launch,
                       "hot_path_conv_gflop_fwd_per_pair": round(flops["total"] / 1e9, 3)},
            "hot_path_only_ms": None if hot_ms is None else round(hot_ms, 3),
            "hot_path_only": None if hot_ms is None else hot_path_record(hot_ms, hot_launch, hot_kernels, flops["total"],
                                                                          args.batch),
            "extra": extra,
            "roofline": roofline, "roofline_all_conv": roofline_all, "roofline_hbm": hbm, "cpu_baseline": cpu, "kernels": kernels,
            "capture_failed": bool(capture_failed),
            "dropin": dropin,
        }
        if pcie is not None:
            out["pcie_inclusive"] = pcie
    else:
        out = None
    if dist_mode and use_graph:
        g_elapsed = graph_phase(args, rank, out, lambda: timed(
            engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=True)), device)
        if g_elapsed is None and rank == 0:
            out["capture_failed"] = True       # the line below is the EAGER iteration (typically ~30 % slower than a replay)
            out["config"]["launch"] = "eager (hipGraph capture with RCCL collectives FAILED on some rank)"
        elif g_elapsed is not None and g_elapsed < elapsed and rank == 0:
            out.update(value=round(global_batch * args.steps / g_elapsed, 2),
                       ms_per_step=round(g_elapsed / args.steps * 1e3, 3))
            out["config"]["launch"] = "hipGraph replay (RCCL collectives captured); eager: %.3f ms/step" % ms_per_step
    if dist_mode:
        # a peer-to-peer SyncBN exchange that gave up waiting for a rank leaves wrong statistics behind: say so, on every rank
        from mnk import dist as mdist
        code = torch.tensor([mdist.p2p_error()], dtype=torch.int32, device=device)
        dist.all_reduce(code, op=dist.ReduceOp.MAX)
        if rank == 0:
            if mdist._P2P["handle"] is not None:
                import ctypes
                from mnk import _lib
                kind = _lib.lib().query("mnk_p2p_memory_kind", ctypes.c_void_p(mdist._P2P["handle"]))
                out["syncbn_exchange"] = "peer-to-peer (csrc/p2p.hip), mailboxes in %s device memory" % (
                    {3: "uncached", 1: "fine-grained", 0: "ordinary"}.get(kind, "?"))
            else:
                out["syncbn_exchange"] = "collective (RCCL / torch.distributed)"
            out["p2p_error"] = int(code.item())
            if p2p_fell_back:
                out["syncbn_exchange"] += " -- after the peer-to-peer exchange gave rank %d up in a first measurement" % (p2p_fell_back - 1)
            if out["p2p_error"]:
                out["capture_failed"] = True
                out["config"]["launch"] += " -- INVALID: a peer-to-peer exchange timed out (rank %d)" % (out["p2p_error"] - 1)
    if rank == 0:
        emit(out)
    if dist_mode:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
