#!/usr/bin/env python3
"""bench.py - train-step views/s (fwd+bwd) of the MI355X rasterizer + LGDWT loss path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c4|c2|c1|c5|tiny]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W

A step = one camera per GPU: activations -> GaussianRasterizer forward -> clamp -> LGDWT loss
(0.8 L1 + 0.2 (1-SSIM) + running-mean-scaled global 2-level DWT + 0.1 patch DWT) -> backward to the
six parameter tensors -> densification statistics -> Adam (N = 1: all of the last three inside the
backward's per-Gaussian kernel, gs_backward_step; N > 1: the same kernel writes the 59 gradient floats per
Gaussian, the statistic increments and a validity flag into ONE exchange buffer -> RCCL reduce-scatter ->
gated Adam on this rank's 1/N of the rows -> all-gather of the parameters; GS_SHARDED_ADAM=0: chunked
all-reduce with the optimizer behind the chunks).  The timed step therefore INCLUDES the
optimizer; the metric's "(fwd+bwd)" is BASELINE.json's wording.  Synthetic "trained-like" Gaussians, NeRF-synthetic-
like orbit cameras, inputs resident in HBM before the timed region.  Cameras are sharded over ranks
and per-GPU work is fixed, so scaling is "weak".

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     for the dominant kernel: algorithmic bytes per launch / mean launch duration measured
               with HIP events on the launch stream (library profiler, include/gsplat.h gs_profile_*);
               for the blend kernels (VALU-bound) additionally the VALU-issue fraction, from the instruction
               counts of the committed SQ-counter profile (labelled with its tag) and this run's duration
  cpu_baseline the CPU oracle ("port") timed on this box's host cores on the same workload: all cores
               (2 warm-up views, median of 5) and one thread (one view)
  reference_lists  the same step on the reference's bounding-square instance lists (tile_cull = 0)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # name: (P, W, H, dwt, patch, description)
    "c3": (1_000_000, 1920, 1080, True, True, "BASELINE configs[2]: 1M Gaussians, 1080p, global+patch DWT"),
    # the per-GPU share of configs[3] (8 cameras / step over 8 GPUs = one camera per GPU per step, full replica)
    "c4": (2_000_000, 1920, 1080, True, True, "BASELINE configs[3]: 2M Gaussians, 1080p, one camera per GPU per step"),
    "c2": (500_000, 800, 800, True, False, "BASELINE configs[1]: 500k Gaussians, 800x800, global DWT"),
    "c1": (10_000, 400, 400, False, False, "BASELINE configs[0]: 10k Gaussians, 400x400, DWT off"),
    "tiny": (2_000, 256, 160, True, True, "plumbing check"),
    # not a BASELINE config: a size check of the buffers and index arithmetic (tests/tools/step_probe.py x4k)
    "x4k": (6_000_000, 3840, 2160, True, True, "stress: 6M Gaussians at 3840x2160"),
    # multispectral step (train_nir.py: L1 + SSIM on RGB and on the NIR image, no DWT terms): ONE fused 4-channel pass
    # as BASELINE.json words it: the 4-channel pass with the LGDWT criterion (global + patch DWT) on the RGB image
    "c5": (1_000_000, 1920, 1080, True, True, "BASELINE configs[4]: RGB+NIR 4-channel render, 1M Gaussians, 1080p, "
                                              "global+patch DWT on the RGB image"),
    # the reference's train_nir.py as written: L1 + SSIM on RGB and NIR, no DWT terms
    "c5_plain": (1_000_000, 1920, 1080, False, False, "RGB+NIR 4-channel render, 1M Gaussians, 1080p, train_nir.py's loss"),
}
NIR_CONFIGS = ("c5", "c5_plain")
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def stage_bytes(P, R, N):
    """Algorithmic (compulsory) bytes per launch of each rasterizer stage, SURVEY.md 8(d):
    per view 902 P + 172 R + 48 N split over the stages that move them."""
    return {
        "preprocess_fwd": 311 * P, "scan": 8 * P, "duplicate": 20 * P + 12 * R, "sort": 24 * R,
        "tile_ranges": 8 * R, "render_fwd": 44 * R + 24 * N, "render_bwd": 84 * R + 24 * N,
        "preprocess_bwd": 563 * P,
        # gs_backward_step: the same stage without its 248 B of gradient writes (307 B read), plus the raw rotation /
        # scaling / opacity rows (32 B), both Adam moments read and written, the parameters written (3 x 236 + 472 B ...)
        "preprocess_bwd_step": (307 + 32 + 2 * 236 + 3 * 236 + 24) * P,
        "adam": 28 * 59 * P,
    }


SCENES = {
    # SURVEY 8(d) "trained-like": anisotropic, random rotations / opacities, full SH - the headline workload
    "trained_like": dict(sh_degree=3, what="trained-like (SURVEY 8d), seed 0"),
    # SURVEY 8(d) "init-like": what create_from_pcd makes of a random point cloud (gaussian_model.py:149-176) - isotropic,
    # opacity 0.1, active SH degree 0 (the reference's iteration-1 state): nothing saturates early
    "init_like": dict(sh_degree=0, what="init-like (SURVEY 8d): isotropic, opacity 0.1, sh_degree 0, seed 0"),
    # half of the image never saturates (gsplat_amd.synthetic.ball_in_shell): depth limits cannot cut those tiles
    "ball_in_shell": dict(sh_degree=3, what="dense ball inside a thin low-opacity shell: about half the tiles never saturate"),
}


# The model keeps its rows in Morton order of the Gaussians' centres (GaussianModelLite.spatial_order - the model's own default from
# 100 000 Gaussians on: a permutation of the same scene, applied at construction and after every densification);
# GS_BENCH_SPATIAL_ORDER=0 / 1 = rows as generated / ordered at every size.
SPATIAL_ORDER = None if os.environ.get("GS_BENCH_SPATIAL_ORDER", "") not in ("0", "1") else os.environ["GS_BENCH_SPATIAL_ORDER"] == "1"


def build_workload(cfg, device, rank, world, seed=0, scene_kind="trained_like", gts=None):
    import diff_gaussian_rasterization as dgr
    import lgdwt_loss
    from gsplat_amd import synthetic
    from gsplat_amd.trainer import GaussianModelLite, Trainer, camera_to, render
    from gsplat_amd._lib import hip_api as hip_api_
    from simple_knn._C import distCUDA2

    P, W, H, dwt, patch, _ = CONFIGS[cfg]
    knn = lambda x: distCUDA2(x.to(device)).cpu()  # noqa: E731
    make_scene = getattr(synthetic, scene_kind)
    scene = make_scene(P, seed=seed, knn=knn, sh_degree=SCENES[scene_kind]["sh_degree"])
    cams = [camera_to(c, device) for c in synthetic.orbit_cameras(W, H)]
    bg = torch.zeros(3, device=device)
    if gts is None:
        # ground truth: renders of a differently seeded scene, quantised to 8 bit like PILtoTorch
        gt_scene = make_scene(P, seed=seed + 1, knn=knn, sh_degree=SCENES[scene_kind]["sh_degree"])
        gt_model = GaussianModelLite(gt_scene, device, api=hip_api_())
        # every rank only ever touches cameras rank, rank+world, ...: render just those
        needed = sorted({(k * world + rank) % len(cams) for k in range(len(cams))})
        gts = [None] * len(cams)
        with torch.no_grad():
            for ci in needed:
                img = render(cams[ci], gt_model, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, bg)["render"]
                gts[ci] = (torch.round(img * 255.0) / 255.0).contiguous()
        del gt_model
    if cfg in NIR_CONFIGS:
        from gsplat_amd.losses import LossOps
        from gsplat_amd.trainer import NirCriterion, TrainerNIR
        g = torch.Generator().manual_seed(seed + 2)
        nirs = [None if x is None else (torch.round(torch.rand((1, H, W), generator=g) * 255.0) / 255.0).to(device) for x in gts]
        model = GaussianModelLite(scene, device, api=hip_api_(), with_nir=True, spatial_order=SPATIAL_ORDER)
        # the multispectral step on the fused machinery (GS_BENCH_NIR_FUSED=0: round-3 form - un-fused criterion and optimizer tail)
        nir_fused = os.environ.get("GS_BENCH_NIR_FUSED", "1") != "0"
        rgb_crit, masks = None, None
        if dwt or patch:
            rgb_crit = lgdwt_loss.criterion(dwt_enable=dwt, patch_dwt_enable=patch, fused=nir_fused)
            if patch:
                masks = [None if x is None else rgb_crit.elf_mask(x) for x in gts]
        tr = TrainerNIR(model, cams, gts, nirs, NirCriterion(LossOps(hip_api_()), rgb_criterion=rgb_crit, fused=nir_fused),
                        dgr.GaussianRasterizationSettings, bg, rank=rank, world_size=world, masks=masks)
        return tr, scene, cams, gts
    model = GaussianModelLite(scene, device, api=hip_api_(), spatial_order=SPATIAL_ORDER)
    crit = lgdwt_loss.criterion(dwt_enable=dwt, patch_dwt_enable=patch)
    masks = None
    if patch:  # ELF / patch selection depends on the ground truth only: cached per camera
        masks = [None if g is None else crit.elf_mask(g) for g in gts]
    tr = Trainer(model, cams, gts, crit, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, bg, rank, world,
                 optimizer_step=True, masks=masks)
    return tr, scene, cams, gts


def host_cpu_share():
    """CPUs this process may really use: min(affinity mask, cgroup quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("GS_CPU_THREADS", n))))


def cpu_baseline(cfg, scene, cam, gt, log, views=5, warmups=2):
    """The CPU oracle (C++ restatement of the reference kernels, OpenMP) on a bounded sample of the same workload,
    protocol of SURVEY 8(d): forward + loss + backward of one camera on the host cores, (i) all cores this process may
    use (OMP threads pinned by OMP_PROC_BIND when the caller sets it), `warmups` untimed views then the MEDIAN of
    `views` timed ones (~15 s at C3), (ii) one thread, one view (~25 s at C3; GS_CPU_BASELINE_1T=0 skips it).
    Reported, never the thing measured above."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import statistics

    import oracle_lib
    from gsplat_amd.losses import LGDWTCriterion, LossOps
    from gsplat_amd.trainer import GaussianModelLite, camera_to, render

    P, W, H, dwt, patch, _ = CONFIGS[cfg]
    orc = oracle_lib.get()
    cpu = torch.device("cpu")
    model = GaussianModelLite({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in scene.items()}, cpu, api=orc.api)
    crit = LGDWTCriterion(LossOps(orc.api), dwt_enable=dwt, patch_dwt_enable=patch)
    cam = camera_to(cam, cpu)
    gt = gt.cpu()
    orc.lib.gso_set_num_threads.restype = int

    def one_view():
        t0 = time.perf_counter()
        model.zero_grad()
        pkg = render(cam, model, orc.Rasterizer, orc.Settings, torch.zeros(3), filter_as_indices=False)
        loss, _ = crit(pkg["render"], gt)
        loss.backward()
        return time.perf_counter() - t0, float(loss.detach())

    cores = int(orc.lib.gso_set_num_threads(host_cpu_share()))
    torch.set_num_threads(cores)
    for _ in range(warmups):
        one_view()
    times = [one_view()[0] for _ in range(views)]
    dt = statistics.median(times)
    log("cpu_baseline: %d threads, %d warm-ups, %d views: median %.2f s (min %.2f, max %.2f)" % (
        cores, warmups, views, dt, min(times), max(times)))
    out = {"value": 1.0 / dt, "unit": "views/s", "cores": cores, "kind": "port",
           "sample": "%d warm-up + %d timed views (median; fwd + loss + bwd each, no Adam) of the same %s workload through "
                     "the CPU oracle (oracle/libgs_oracle.so, OpenMP over Gaussians / tiles)" % (warmups, views, cfg),
           "seconds_per_view": {"median": dt, "min": min(times), "max": max(times)}}
    if os.environ.get("GS_CPU_BASELINE_1T", "1") != "0":
        orc.lib.gso_set_num_threads(1)
        torch.set_num_threads(1)
        t1, _ = one_view()
        log("cpu_baseline: 1 thread, 1 view: %.2f s" % t1)
        out["one_thread"] = {"value": 1.0 / t1, "unit": "views/s", "cores": 1, "sample": "1 view, no warm-up"}
        orc.lib.gso_set_num_threads(cores)
        torch.set_num_threads(cores)
    return out


# wave64 VALU issue ceiling, MEASURED (tests/tools/valu_peak_probe.hip, profiles/r03_valu_peak_probe.json): independent
# v_fma_f32 streams retire 1 105 G wave-instructions/s at 4 resident waves per SIMD (1 152 at 8; 980 at 3; 511 with one
# wave alone) = 0.90-0.94 of the nominal 1 024 SIMDs x 2.4 GHz / 2 cycles = 1 228.8.  The 2-cycle rate is real once two
# waves share a SIMD - the 4-cycle price of MI355X_MICROARCH.md holds for ONE wave alone (round 2 re-priced with it:
# withdrawn).  v_pk_fma_f32 / v_pk_mul_f32 retire at HALF that rate (565 G/s: two lanes of work per instruction, same
# flops), so packing buys nothing; a transcendental (v_exp_f32, v_rcp_f32) costs about 7 plain instructions: a stream
# shaped like the alpha test (11 plain + 1 v_exp_f32) tops out at 620-633 G/s.
VALU_PEAK_NOMINAL_GINST = 1024 * 2.4 / 2.0


def measured_valu_ceilings():
    """{"fma": G wave-inst/s of plain f32 ops at 4 waves/SIMD, "blendmix": of the 11:1 plain:transcendental stream}"""
    out = {"fma": 1105.4, "blendmix": 620.6, "from": "constants (profiles/r03_valu_peak_probe.json missing)"}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r03_valu_peak_probe.json")))
        for r in d["results"]:
            if r["waves_per_simd"] == 4 and r["kind"] in ("fma", "blendmix"):
                out[r["kind"]] = float(r["g_wave_inst_per_s"])
        out["from"] = "profiles/r03_valu_peak_probe.json (tests/tools/valu_peak_probe.hip on an MI355X box, 4 waves/SIMD)"
    except Exception:
        pass
    return out


def counters_match(d, P, R):
    """May the static counter file `d` (profiles/pmc_traffic.json / sq_insts.json) be combined with THIS run's durations?
    -> None when yes, else the reason: other kernel sources (profiles/source_id.py), another P, or an instance count R
    that differs by more than 5 % from the profiled run's."""
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    try:
        from source_id import source_id
        sid = source_id(ROOT)
    except Exception as e:   # noqa: BLE001
        return "source id unavailable (%r)" % (e,)
    if d.get("source_id") != sid:
        return "counters were collected from other kernel sources (file: %s, this build: %s)" % (d.get("source_id"), sid)
    if P is not None and d.get("P") not in (None, P):
        return "counters were collected with %s Gaussians, this run has %s" % (d.get("P"), P)
    if R is not None and d.get("R"):
        if abs(float(R) - float(d["R"])) > 0.05 * float(d["R"]):
            return "counters were collected at R = %s instances per view, this run has %s" % (d["R"], R)
    return None


def valu_roofline(kernel, ms_per_launch, P=None, R=None, counters="sq_insts.json"):
    """VALU-issue fraction of a blend kernel: wave-instructions per launch (SQ_INSTS_VALU of the committed SQ-counter
    profile, profiles/sq_insts.json: {"tag": ..., "<kernel>": insts per launch}) / this run's launch duration / the
    MEASURED plain-f32 issue ceiling."""
    f = os.path.join(ROOT, "profiles", counters)
    try:
        d = json.load(open(f))
        insts = float(d[kernel])
    except Exception:
        return None
    why = counters_match(d, P, R)
    if why is not None:
        return {"bound": "valu", "achieved": None, "frac": None, "dropped": why}
    ceil = measured_valu_ceilings()
    ach = insts / (ms_per_launch * 1e-3) / 1e9
    # `frac` / `peak` = against the NOMINAL issue peak (1 024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction); the measured
    # ceiling of a plain-f32 stream (0.90 of it at 4 waves per SIMD) beside it
    return {"bound": "valu", "achieved": ach, "peak": VALU_PEAK_NOMINAL_GINST, "unit": "G wave-instructions/s",
            "frac": ach / VALU_PEAK_NOMINAL_GINST, "insts_per_launch": insts,
            "peak_from": "nominal: 1024 SIMDs x 2.4 GHz / 2 cycles (MI355X_MICROARCH.md)",
            "peak_measured": ceil["fma"], "peak_measured_from": ceil["from"],
            "frac_of_measured_ceiling": ach / ceil["fma"],
            "mix_ceiling": ceil["blendmix"],
            "mix_ceiling_note": "measured rate of a stream of 11 plain f32 ops per v_exp_f32 (the alpha test's shape): a "
                                "transcendental costs ~7 plain issue slots, so a kernel with them cannot reach `peak`",
            "insts_from": "profiles/%s (tag %s): SQ_INSTS_VALU per launch on the %s; duration measured "
                          "in this run" % (counters, d.get("tag", "?"), d.get("scene", "C3 workload"))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timers", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product path has no CPU fallback"
    # rehearsal knobs (control-flow check of the N>1 path on a ONE-GPU box): all ranks on device 0 over gloo
    one_dev = os.environ.get("GS_BENCH_SINGLE_DEVICE", "0") == "1"
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    from gsplat_amd._lib import hip_api
    from gsplat_amd.capi import read_profile
    api = hip_api()
    P, W, H, dwt, patch, desc = CONFIGS[args.config]
    t_setup = time.perf_counter()
    # (developer switch for profiles of SURVEY 8d's other inputs - profiles/collect.sh: the driver's line is the default scene)
    main_scene = os.environ.get("GS_BENCH_SCENE", "trained_like")
    tr, scene, cams, gts = build_workload(args.config, device, rank, world, scene_kind=main_scene)
    torch.cuda.synchronize()
    log("workload %s built in %.1f s" % (args.config, time.perf_counter() - t_setup))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # single GPU: the steady-state step is replayed from a hipGraph (gsplat_amd.trainer.GraphedStep: same kernels and
    # arguments as the eager step, one graph launch instead of ~45 kernel launches with their Python glue);
    # GS_BENCH_GRAPH=0 times the eager step.  The event timers need eager launches: stage passes run eagerly.
    # GS_BENCH_GRAPH: 0 (default since round 5) = the eager step - since the step left the autograd engine (round 4) the replay
    # no longer wins at C1 or C3 (0.349 vs 0.326 ms, 0.955 vs 0.867 ms: profiles/r04_bench_c*.json), only at C2 - so the bench
    # no longer picks a form by trial; auto = both forms are timed over a few untimed steps and the faster one runs the
    # timed region; 1 = the replay.
    graph_mode = os.environ.get("GS_BENCH_GRAPH", "0")
    nir_unfused = args.config in NIR_CONFIGS and os.environ.get("GS_BENCH_NIR_FUSED", "1") == "0"
    use_graph = world == 1 and not nir_unfused and graph_mode != "0"
    graphed, graph_choice = None, None
    if use_graph:
        from gsplat_amd.trainer import GraphedStep
        graphed = GraphedStep(tr)

    # depth-limited instance lists (csrc/gs_tilecull.h; exact, verified by the forward, verdict collected one step later -
    # gsplat_amd.trainer.Trainer.depth_limit): GS_BENCH_DEPTH_LIMIT=0 switches them off
    # (default: on from 100 k Gaussians - on the 10 k scene of c1 there is nothing to cut and the two extra launches cost 3 %)
    depth_limit = not nir_unfused and \
        os.environ.get("GS_BENCH_DEPTH_LIMIT", "1" if P >= 100_000 else "0") != "0"
    if depth_limit:
        tr.depth_limit = "deferred"

    def run_step(kk):
        return graphed.step(kk) if graphed is not None else tr.step(kk)

    def settle():
        if graphed is not None:
            graphed.sync()
        elif hasattr(tr, "sync"):
            tr.sync()

    k = 0
    for _ in range(args.warmup):
        tr.step(k)
        k += 1
    barrier()
    # HIP-event timers (library profiler).  Every event pair drains the pipeline for ~10 us, so the timed region carries
    # them around ONE kernel only - the dominant one, found by an untimed pass with every stage instrumented first.
    names = [api.raw("profile_stage_name")(i).decode() for i in range(api.raw("profile_stage_count")())]
    prof, prof_timed, dom_stage = {}, {}, os.environ.get("GS_BENCH_DOMINANT", "")
    prof_alone = None

    def profile_all_stages(nsteps, trainer=None, k0=None):
        nonlocal k
        t_ = trainer if trainer is not None else tr
        api.call("profile_reset")
        api.call("profile_only", -1)
        api.call("profile_enable", 1)
        for j in range(nsteps):
            if k0 is None:
                t_.step(k)
                k += 1
            else:
                t_.step(k0 + j)
        t_.sync()
        barrier()
        api.call("profile_enable", 0)
        return read_profile(api)

    def stage_ms(pr):
        pr = dict(pr)
        if "sort_depth" in pr and "sort" in pr:
            pr["sort"] = (pr["sort"][0] + pr.pop("sort_depth")[0], pr["sort"][1])
        return {n: round(ms / cnt, 5) for n, (ms, cnt) in pr.items()}

    if not args.no_stage_timers and not dom_stage:
        pre = profile_all_stages(min(args.steps, 4))
        timed_kernels = [n for n in pre if n in stage_bytes(1, 1, 1)]
        dom_stage = max(timed_kernels, key=lambda n: pre[n][0] / pre[n][1]) if timed_kernels else "render_bwd"
    # steady state = every camera has been visited before (sparse-view training revisits its few cameras all the time):
    # one more untimed cycle over the camera set, which also fills the forward's per-camera tile-order hints
    cycle = len(cams) // world + 2
    trial = {}
    for _ in range(cycle):
        tr.step(k)
        k += 1
    barrier()
    # N > 1: which form of the gradient exchange runs the timed region - the dense sharded one (reduce-scatter, Adam on 1/N,
    # all-gather: 244 B per Gaussian whatever the views saw) or the visibility-sparse one (only the rows of Gaussians some
    # rank's view emitted instances for; one host read per step).  Which is faster depends on the scene and on the links:
    # both are timed over a few untimed steps (GS_BENCH_EXCHANGE = auto, default), every rank takes the slower rank's
    # times and so the same decision; dense / sparse force a form.  GS_SPARSE_EXCHANGE=1 (Trainer's own switch) = sparse.
    exchange_trial = None
    ex_mode = os.environ.get("GS_BENCH_EXCHANGE", "auto")
    if world > 1 and hasattr(tr, "sparse_exchange") and not tr.sparse_exchange and ex_mode in ("auto", "sparse"):
        def time_form(sparse, n=6):
            nonlocal k
            tr.sparse_exchange = sparse
            for _ in range(2):
                tr.step(k)
                k += 1
            tr.sync()
            barrier()
            t1 = time.perf_counter()
            for _ in range(n):
                tr.step(k)
                k += 1
            tr.sync()
            barrier()
            tt = torch.tensor([(time.perf_counter() - t1) / n * 1e3], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt[0])
        if ex_mode == "auto":
            # sparse first: it keeps every replica's moments whole, the sharded form after it may let the other shards' go stale
            exchange_trial = {"sparse_ms_per_step": time_form(True), "dense_sharded_ms_per_step": time_form(False)}
            exchange_trial["chosen"] = "sparse" if exchange_trial["sparse_ms_per_step"] < exchange_trial["dense_sharded_ms_per_step"] \
                else "dense"
            log("exchange form: sparse %.3f ms, dense %.3f ms per step -> %s" % (
                exchange_trial["sparse_ms_per_step"], exchange_trial["dense_sharded_ms_per_step"], exchange_trial["chosen"]))
        if ex_mode == "sparse" or exchange_trial["chosen"] == "sparse":
            tr.sparse_exchange = False
            tr.gather_optimizer_state()     # (the sharded steps of the trial left each rank with current moments of its shard only)
            tr.sparse_exchange = True
        else:
            tr.sparse_exchange = False
        for _ in range(2):
            tr.step(k)
            k += 1
        tr.sync()
        barrier()
    if graphed is not None:
        if graph_mode == "auto":
            t1 = time.perf_counter()
            for _ in range(8):
                tr.step(k)
                k += 1
            barrier()
            trial["eager"] = (time.perf_counter() - t1) / 8 * 1e3
        # the first of these steps captures the graph (three warm-up steps on a side stream + the captured one)
        for _ in range(cycle):
            graphed.step(k)
            k += 1
        graphed.sync()
        barrier()
        if graph_mode == "auto":
            t1 = time.perf_counter()
            for _ in range(8):
                graphed.step(k)
                k += 1
            graphed.sync()
            barrier()
            trial["graph"] = (time.perf_counter() - t1) / 8 * 1e3
            graph_choice = {"graph_ms_per_step": trial["graph"], "eager_ms_per_step": trial["eager"],
                            "chosen": "graph" if trial["graph"] < trial["eager"] else "eager"}
            log("launch form: graph %.3f ms, eager %.3f ms per step -> %s" % (trial["graph"], trial["eager"], graph_choice["chosen"]))
            if graph_choice["chosen"] == "eager":
                graphed.sync()
                graphed = None
                for _ in range(3):  # back on the eager allocator state before timing
                    tr.step(k)
                    k += 1
                barrier()
    from gsplat_amd import hip_backend as _hb0
    # (no cyclic-garbage collection inside the 30 ms timed window: a generation-2 sweep of the interpreter is milliseconds)
    import gc
    gc.collect()
    gc.disable()
    # the sweep above (and the set-up before it) left the GPU idle for tens of milliseconds: a few more untimed steps so that
    # the timed region starts on a busy device, as every step of a training run does
    for _ in range(3):
        run_step(k)
        k += 1
    settle()
    if graphed is None and not args.no_stage_timers:  # an eager timed region carries the event pair of the dominant kernel
        api.call("profile_reset")
        api.call("profile_only", names.index(dom_stage))
        api.call("profile_enable", 1)
    if world > 1:
        tr.exchange_events = []   # (Trainer.exchange_and_step brackets the exchange + optimizer with HIP events)
    barrier()
    dl0 = dict(_hb0().depth_limit_stats)
    t0 = time.perf_counter()
    sampled = graphed is None and not args.no_stage_timers
    for i in range(args.steps):
        if sampled:  # the dominant kernel's event pair (~10 us of pipeline drain) rides on every 4th step of the region
            api.call("profile_enable", 1 if i % 4 == 0 else 0)
        run_step(k)
        k += 1
    settle()  # (the last step's deferred verdict: inside the timed region)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    # the exchange brackets of the TIMED steps only (the stage-timer passes and the other legs below step the model too)
    timed_exchange_events = []
    if world > 1:
        timed_exchange_events, tr.exchange_events = list(tr.exchange_events or []), None
    dl1 = dict(_hb0().depth_limit_stats)
    R_timed = int(_hb0()._pinned[0]) if _hb0()._pinned is not None else 0
    # how many 256-row blocks of the model are dormant (all Adam moments +0: FlatAdam.dormant_flags) after the timed region
    dormant_info = None
    opt_ = getattr(tr.model, "optimizer", None)
    if world == 1 and getattr(opt_, "USE_DORMANT", False) and hasattr(opt_, "dormant_flags"):
        fl = opt_.dormant_flags()
        dormant_info = {"blocks": int(fl.numel()), "dormant": int(fl.sum())}
    if depth_limit and _hb0().last_deferred_num_rendered is not None:  # (deferred forwards report their count with the verdict)
        R_timed = int(_hb0().last_deferred_num_rendered)
    if not args.no_stage_timers:
        if graphed is None:
            api.call("profile_enable", 0)
            prof_timed = read_profile(api)
            api.call("profile_only", -1)
        # every stage, in an untimed eager pass of the same steady-state step right after the timed region
        prof = profile_all_stages(min(args.steps, 10))
        if dom_stage in prof_timed:
            prof[dom_stage] = prof_timed[dom_stage]  # the timed region's own measurement of the dominant kernel
        # the two-phase step runs the Adam stream of the Gaussians without instances BESIDE the backward blend: both
        # kernels' durations above are those of the pair.  The same step with every kernel on its own (one launch for the
        # whole per-Gaussian stage), untimed, for the per-kernel rooflines
        from gsplat_amd import hip_backend as _hb1
        if world == 1 and _hb1().two_phase_launches > 0:
            _hb1().TWO_PHASE = False
            try:
                p1 = profile_all_stages(min(args.steps, 6))
            finally:
                del _hb1().TWO_PHASE
            prof_alone = {k_: ms_ / cnt_ for k_, (ms_, cnt_) in p1.items()}
    # the same step on the reference's bounding-square instance lists (GsView.tile_cull = 0: point_list / ranges /
    # num_rendered bit-identical to the reference's), outside the timed region
    ref_lists = None
    if world == 1 and os.environ.get("GS_BENCH_REFERENCE_LISTS", "1") != "0":
        from gsplat_amd import hip_backend as _hb
        be = _hb()
        old_cull = be.tile_cull
        be.tile_cull = False
        try:
            for _ in range(2):
                tr.step(k)
                k += 1
            barrier()
            nref = min(args.steps, 10)
            t1 = time.perf_counter()
            for _ in range(nref):
                tr.step(k)
                k += 1
            barrier()
            dref = (time.perf_counter() - t1) / nref
            ref_stages = None if args.no_stage_timers else stage_ms(profile_all_stages(min(nref, 6)))
            ref_lists = {"ms_per_step": dref * 1e3, "views_per_s": 1.0 / dref, "steps": nref, "stages_ms_per_launch": ref_stages,
                         "num_rendered_last_view": int(be._pinned[0]) if be._pinned is not None else None,
                         "what": "the identical step with GsView.tile_cull = 0: the instance lists (num_rendered, point_list, "
                                 "ranges) are the reference's, bit for bit; untimed by the driver"}
        finally:
            be.tile_cull = old_cull
        for _ in range(2):  # restore the capacity hint / last-view counter of the culled mode
            tr.step(k)
            k += 1
        barrier()
    # Two options of the reference's loop on the same step, untimed by the driver (verdict r04 items 3 and 5):
    #  * optimizer_type = "sparse_adam" (train.py:282-284): only the Gaussians visible in the view are stepped - the Adam stream
    #    beside the backward blend shrinks to the visible Gaussians without instances;
    #  * the depth-regularisation term (train.py:204-216) with a synthetic monocular prior per camera.
    options = None
    if world == 1 and args.config not in NIR_CONFIGS and os.environ.get("GS_BENCH_OPTIONS", "1") != "0":
        options = {}
        nopt = min(args.steps, 10)

        def timed_steps(n):
            nonlocal k
            for _ in range(3):
                tr.step(k)
                k += 1
            tr.sync()
            barrier()
            t1 = time.perf_counter()
            for _ in range(n):
                tr.step(k)
                k += 1
            tr.sync()
            barrier()
            return (time.perf_counter() - t1) / n * 1e3
        opt_ = tr.model.optimizer
        if hasattr(opt_, "sparse"):
            opt_.sparse = True
            try:
                ms = timed_steps(nopt)
                st = None if args.no_stage_timers else stage_ms(profile_all_stages(min(nopt, 6)))
                options["sparse_adam"] = {"ms_per_step": ms, "views_per_s": 1e3 / ms, "stages_ms_per_launch": st,
                                          "what": "Trainer(optimizer_type='sparse_adam'): GsStepState.sparse = 1 - Gaussians with "
                                                  "radii <= 0 keep parameters and moments; same step otherwise"}
            finally:
                opt_.sparse = False
            if hasattr(opt_, "invalidate_dormant"):
                opt_.invalidate_dormant()
        g_ = torch.Generator().manual_seed(11)
        tr.depth_priors = [((torch.rand((1, H, W), generator=g_) * 0.5).to(device), None) for _ in cams]
        tr.depth_l1_weight = 0.5
        try:
            ms = timed_steps(nopt)
            st = None if args.no_stage_timers else stage_ms(profile_all_stages(min(nopt, 6)))
            options["depth_regularisation"] = {"ms_per_step": ms, "views_per_s": 1e3 / ms, "stages_ms_per_launch": st,
                                               "what": "the same step + depth_l1_weight x mean |invDepth - mono_invdepth| (one "
                                                       "gs_depth_l1 launch: term and image gradient; the blend backward then also "
                                                       "carries the inverse-depth channel)"}
        finally:
            tr.depth_priors, tr.depth_l1_weight = None, 0.0
        for _ in range(2):
            tr.step(k)
            k += 1
        barrier()
    # The DROP-IN loop, untimed by the driver: the reference's own iteration (LGDWT-GS/train.py:97-288) written against the
    # drop-in packages only - GaussianRasterizer, lgdwt_loss.l1_loss / get_dwt_subbands / compute_elf_map /
    # compute_patch_dwt_loss, fused_ssim, torch.optim.Adam over six tensors, the reference's `.item()` syncs - with the
    # product's defaults (no environment switches): what a maintainer who follows INTEGRATION.md section 1 gets
    # (gsplat_amd/dropin.py).  Then the two one-line additions INTEGRATION.md section 4 offers: the optimizer as one
    # kernel per tensor (gsplat_amd.optim.FusedAdam, same constructor) and a camera_key on the rasterizer (depth-limited
    # lists on a camera's later visits, verified by the forward).  Last: section 5's `gsplat_amd.render_raw.render` in place of
    # the reference's render() (the model's six raw tensors go to the library, SH rows read where the model keeps them) with
    # the criterion as one node on its un-clamped image (criterion.fused_call).  loss.item() sits where train.py:224 has it.
    drop_in = None
    if world == 1 and args.config not in NIR_CONFIGS and os.environ.get("GS_BENCH_DROP_IN", "1") != "0":
        from gsplat_amd.dropin import DropInLoop
        drop_in = {"what": "LGDWT-GS/train.py:97-288 against the drop-in packages only (gsplat_amd/dropin.py): "
                           "GaussianRasterizer + lgdwt_loss functions + fused_ssim + Adam over six tensors, the reference's "
                           "host syncs kept; product defaults, no environment switches; untimed by the driver"}
        n_di = min(args.steps, 20)
        for label, kw in (("torch.optim.Adam", dict(optimizer="torch")),
                          ("FusedAdam", dict(optimizer="fused")),
                          ("FusedAdam + camera_key", dict(optimizer="fused", use_camera_key=True)),
                          ("FusedAdam + camera_key + lgdwt_loss.criterion()", dict(optimizer="fused", use_camera_key=True,
                                                                                 fused_criterion=True)),
                          ("render_raw + FusedAdam + camera_key + lgdwt_loss.criterion()",
                           dict(optimizer="fused", use_camera_key=True, fused_criterion=True, raw_render=True))):
            from gsplat_amd import hip_backend as _hbd
            bed = _hbd()
            loop = DropInLoop(scene, cams, gts, device, dwt=dwt, patch=patch, **kw)
            for j in range(len(cams) + 2):      # every camera once (a keyed camera's limits exist from its second visit on)
                loop.iteration(j % len(cams))
            torch.cuda.synchronize()
            dd0 = dict(bed.depth_limit_stats)
            # (as in the timed region: no cyclic-garbage sweep inside a 60 ms window - a generation-2 sweep of this process is
            #  tens of milliseconds and used to land in one variant or another: 6.5-8.5 ms/step instead of 3.1)
            gc.collect()
            gc.disable()
            t1 = time.perf_counter()
            for j in range(n_di):
                loop.iteration((j + 2) % len(cams))
            torch.cuda.synchronize()
            gc.enable()
            drop_in[label] = {"ms_per_step": (time.perf_counter() - t1) / n_di * 1e3, "steps": n_di,
                              "depth_limited_views": bed.depth_limit_stats["used"] - dd0["used"],
                              "fallbacks": bed.depth_limit_stats["failed"] - dd0["failed"]}
            log("drop-in loop, %s: %.3f ms/step" % (label, drop_in[label]["ms_per_step"]))
            del loop
        drop_in["drop_in_api_ms_per_step"] = drop_in["torch.optim.Adam"]["ms_per_step"]
        torch.cuda.empty_cache()
        for _ in range(2):  # restore the capacity hint / last-view counter of the timed mode
            tr.step(k)
            k += 1
        barrier()
    # The LONG run, untimed by the driver: the timed region is twenty steps early in a run; what a few hundred more steps cost
    # depends on how fast the model moves between two visits of a camera - the depth limits of a camera are the stop depths of
    # its last visit, and a view whose tiles now saturate deeper is rendered again with full lists (a fall-back).  Two runs of
    # GS_BENCH_SUSTAINED (default 600) eager steps each, continuing the timed model: (a) this bench's target - renders of an
    # UNRELATED random scene, which the model can never fit: it keeps moving fast - and (b) a target the model is close to
    # (the same scene with slightly different colours: a run that is converging, where training spends its time).
    sustained = None
    n_sus = int(os.environ.get("GS_BENCH_SUSTAINED", "600"))
    if world == 1 and args.config == "c3" and depth_limit and n_sus > 0 and os.environ.get("GS_BENCH_OTHER_SCENES", "1") != "0":
        from gsplat_amd import hip_backend as _hb3
        be3 = _hb3()
        sustained = {}

        def run_sustained(trainer, label, what):
            kk = 10_000
            for _ in range(len(cams) + 2):
                trainer.step(kk)
                kk += 1
            trainer.sync()
            barrier()
            d0 = dict(be3.depth_limit_stats)
            t1 = time.perf_counter()
            for _ in range(n_sus):
                trainer.step(kk)
                kk += 1
            trainer.sync()
            barrier()
            dx = (time.perf_counter() - t1) / n_sus
            d1 = dict(be3.depth_limit_stats)
            levels = [0] * len(be3.SLACK)
            for ci in range(len(cams)):
                ent = be3.camera_entry(W, H, camera_key=("trainer", trainer.uid, ci), device_index=device.index)
                if ent is not None:
                    levels[ent["slack_level"]] += 1
            sustained[label] = {"target": what, "steps": n_sus, "ms_per_step": dx * 1e3, "views_per_s": 1.0 / dx,
                                "limited_views": d1["used"] - d0["used"], "fallbacks": d1["failed"] - d0["failed"],
                                "cameras_per_slack_factor": dict(zip(["%.2f" % f for f in be3.SLACK], levels))}
            log("sustained (%s): %.3f ms/step over %d steps, %d fall-backs" % (label, dx * 1e3, n_sus, d1["failed"] - d0["failed"]))

        run_sustained(tr, "bench_target", "renders of an unrelated random scene (seed + 1): the model cannot fit it and keeps moving")
        from gsplat_amd import synthetic as _syn
        from gsplat_amd.trainer import GaussianModelLite as _GML, render as _render
        from gsplat_amd._lib import hip_api as _hip_api
        import diff_gaussian_rasterization as _dgr
        g_near = torch.Generator().manual_seed(5)
        near = dict(scene, shs=scene["shs"] + 0.02 * torch.randn(scene["shs"].shape, generator=g_near))
        nm = _GML(near, device, api=_hip_api())
        with torch.no_grad():
            gts_near = [_render(c, nm, _dgr.GaussianRasterizer, _dgr.GaussianRasterizationSettings, torch.zeros(3, device=device))
                        ["render"].clone() for c in cams]
        del nm
        tr3, _, _, _ = build_workload(args.config, device, rank, world, gts=gts_near)
        tr3.depth_limit = "deferred"
        run_sustained(tr3, "near_target", "renders of the same scene with slightly different colours: a run that is converging")
        del tr3, gts_near

    # SURVEY 8(d)'s other inputs, untimed by the driver: the same step (same launch form as the timed region's choice is
    # not used here: eager, depth limits as in the timed region) on an init-like scene at SH degree 0 and on a scene
    # whose background never saturates - what the depth limits / early termination cannot shortcut
    other = {}
    if world == 1 and args.config == "c3" and os.environ.get("GS_BENCH_OTHER_SCENES", "1") != "0":
        from gsplat_amd import hip_backend as _hb2
        for kind in ("init_like", "ball_in_shell"):
            t_b = time.perf_counter()
            tr2, _, _, _ = build_workload(args.config, device, rank, world, scene_kind=kind, gts=gts)
            if depth_limit:
                tr2.depth_limit = "deferred"
            be2 = _hb2()
            saved_hints = (be2._capacity_hint, be2._capacity_hint_limited)
            be2._capacity_hint = be2._capacity_hint_limited = 0   # this scene sizes its own binning buffers
            kk = 0
            for _ in range(len(cams) + 2):
                tr2.step(kk)
                kk += 1
            tr2.sync()
            barrier()
            d0 = dict(be2.depth_limit_stats)
            nx = min(args.steps, 10)
            t1 = time.perf_counter()
            for _ in range(nx):
                tr2.step(kk)
                kk += 1
            tr2.sync()
            barrier()
            dx = (time.perf_counter() - t1) / nx
            d1 = dict(be2.depth_limit_stats)
            e = {"scene": SCENES[kind]["what"], "sh_degree": SCENES[kind]["sh_degree"], "ms_per_step": dx * 1e3,
                 "views_per_s": 1.0 / dx, "steps": nx,
                 "num_rendered_last_view": int(be2.last_deferred_num_rendered if (depth_limit and be2.last_deferred_num_rendered
                                                                                   is not None) else be2._pinned[0]),
                 "limited_views": d1["used"] - d0["used"], "fallbacks": d1["failed"] - d0["failed"]}
            if not args.no_stage_timers:   # HIP-event stage times of the same steps (untimed pass, every kernel bracketed)
                e["stages_ms_per_launch"] = stage_ms(profile_all_stages(min(nx, 6), tr2, kk))
                kk += min(nx, 6)
                # (the blend kernels against the measured VALU issue ceiling, where profiles/ holds this scene's counts)
                vf = {kn: valu_roofline(kn, e["stages_ms_per_launch"][kn], P, e["num_rendered_last_view"],
                                        counters="sq_insts_%s.json" % kind)
                      for kn in ("render_bwd", "render_fwd") if kn in e["stages_ms_per_launch"]}
                if any(v is not None for v in vf.values()):
                    e["blend_kernels_valu"] = vf
            if depth_limit:   # and with the full (culled) lists
                tr2.depth_limit = None
                for _ in range(3):
                    tr2.step(kk)
                    kk += 1
                barrier()
                t1 = time.perf_counter()
                for _ in range(nx):
                    tr2.step(kk)
                    kk += 1
                barrier()
                e["ms_per_step_full_lists"] = (time.perf_counter() - t1) / nx * 1e3
                e["num_rendered_full_lists"] = int(be2._pinned[0])
            other[kind] = e
            log("scene %s: %.3f ms/step (%d instances; full lists %s ms), %d limited views, %d fall-backs; leg took %.1f s" % (
                kind, e["ms_per_step"], e["num_rendered_last_view"], "%.3f" % e.get("ms_per_step_full_lists", float("nan")),
                e["limited_views"], e["fallbacks"], time.perf_counter() - t_b))
            del tr2
            be2._capacity_hint, be2._capacity_hint_limited = saved_hints
            torch.cuda.empty_cache()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dp_info = None
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # per rank: the step's wall time, of which exchange (collectives + optimizer, as the compute stream sees them)
        ev = timed_exchange_events
        ex_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))
        mine = torch.tensor([dt / args.steps * 1e3, ex_ms], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dense_bytes = tr.model.exchange.numel() * 4
        lx = getattr(tr, "last_exchange", None)      # visibility-sparse exchange (GS_SPARSE_EXCHANGE=1): what really moved
        nbytes = int(lx["sparse_bytes"]) if lx else dense_bytes
        worst = max(float(x[1]) for x in allr)
        dp_info = {"per_rank_step_ms": [float(x[0]) for x in allr],
                   "per_rank_exchange_ms": [float(x[1]) for x in allr],
                   "per_rank_compute_ms": [float(x[0]) - float(x[1]) for x in allr],
                   # what of the exchange the step still WAITS for: the HIP-event span from the end of the backward to the end
                   # of the optimizer on the compute stream (the mask exchange of the sparse form runs on a side stream behind
                   # the forward and is not in it), slowest rank
                   "exchange_exposed_ms": worst,
                   "exchange_form": ("sparse (union of the ranks' instanced Gaussians: mask all-reduce on a side stream behind the "
                                     "forward, one pack kernel, all-reduce of the union's rows, one unpack kernel; gated Adam in two parts - the rows outside the union under the collective, the union's rows behind it)" if lx else
                                     ("sharded (reduce-scatter, Adam on 1/N, all-gather)" if tr.sharded_optimizer else
                                      "chunked all-reduce, Adam behind the chunks")),
                   "exchange_trial": exchange_trial,
                   "exchange_bytes_per_gpu": nbytes,
                   "exchange_bytes_per_gpu_dense": dense_bytes,
                   "exchange_bytes_per_gpu_sparse": None if not lx else int(lx["sparse_bytes"]),
                   "union_rows_last_step": None if not lx else int(lx["union_rows"]),
                   "rows": int(tr.model.P),
                   # ring / RS+AG traffic per GPU = 2 (N-1)/N x bytes, over the time the slowest rank spent in the exchange
                   "bus_GBps": (2.0 * (world - 1) / world * nbytes / 1e9) / (worst * 1e-3) if worst > 0 else None,
                   "what": "exchange = reduce-scatter + Adam on 1/N + all-gather (sharded) or chunked all-reduce with Adam "
                           "behind the chunks, bracketed by HIP events on the compute stream; compute = step - exchange"}
    dt = float(tmax[0])

    if rank == 0:
        # instance count of the timed cameras (R drives the cost, not P)
        Rs = []
        import diff_gaussian_rasterization as dgr  # noqa: F401
        from gsplat_amd import hip_backend
        R_last = R_timed
        N = W * H
        views = args.steps * world
        value = views / dt
        sb = stage_bytes(P, R_last, N)
        if prof_alone is not None:
            # two-phase step: the per-Gaussian stage's bytes are split between two launches (how, depends on the view): no
            # per-launch byte figure for either; the one-launch form is priced in stages_one_launch_step
            sb.pop("preprocess_bwd_step", None)
        stages = {}
        if "sort_depth" in prof and "sort" in prof:  # the two halves of the (tile|depth) sort: one SURVEY stage
            prof["sort"] = (prof["sort"][0] + prof.pop("sort_depth")[0], prof["sort"][1])
        for name, (ms, cnt) in prof.items():
            e = {"ms_per_launch": ms / cnt, "launches": cnt}
            if name in sb:
                e["algorithmic_GB"] = sb[name] / 1e9
                e["GBps"] = sb[name] / 1e9 / (ms / cnt / 1e3)
            stages[name] = e
        roofline = None
        if stages:
            dom = dom_stage if dom_stage in stages and dom_stage in sb else \
                max((n for n in stages if n in sb), key=lambda n: stages[n]["ms_per_launch"])
            ach = stages[dom]["GBps"]
            hbm = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
            roofline = dict(hbm)
            roofline.update({"kernel": dom, "traffic": None, "algorithmic_bytes_per_launch": sb[dom],
                             "ms_per_launch": stages[dom]["ms_per_launch"],
                             "step_algorithmic_GB_per_view": (902 * P + 172 * R_last + 84 * N) / 1e9,
                             "step_algorithmic_GBps": (902 * P + 172 * R_last + 84 * N) / 1e9 * value / world,
                             # SURVEY 8(d)'s byte model ends at the gradients; the timed step also holds the optimizer:
                             # parameters and both moments of all 59 floats read and written (the fused step never writes
                             # or re-reads the gradients): + 24 B x 59 x P
                             "step_algorithmic_GB_per_view_with_optimizer": (902 * P + 172 * R_last + 84 * N + 24 * 59 * P) / 1e9,
                             "step_algorithmic_GBps_with_optimizer":
                                 (902 * P + 172 * R_last + 84 * N + 24 * 59 * P) / 1e9 * value / world})
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tf) and args.config == "c3":  # counters were collected on the C3 workload
                try:
                    t = json.load(open(tf))
                    why = counters_match(t, P, R_last)
                    if why is None:
                        roofline["traffic"] = t.get(dom)
                        roofline["traffic_from"] = "profiles/pmc_traffic.json (tag %s, same kernel sources, R within 5 %%): " \
                                                   "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench, not " \
                                                   "measured in this run" % t.get("tag", "r01")
                    else:
                        roofline["traffic_dropped"] = why
                except Exception:
                    pass
            # the blend kernels are bound by VALU issue, not by HBM: report that roof for them
            blend = {}
            for kname in ("render_bwd", "render_fwd"):
                if kname in stages and args.config == "c3":  # (the instruction counts are those of the C3 workload)
                    v = valu_roofline(kname, stages[kname]["ms_per_launch"], P, R_last)
                    if v:
                        if prof_alone and kname in prof_alone and v.get("frac") is not None:
                            va = valu_roofline(kname, prof_alone[kname], P, R_last)
                            v["alone"] = {"ms_per_launch": prof_alone[kname], "achieved": va["achieved"], "frac": va["frac"],
                                          "what": "the kernel without the Adam stream of the two-phase step beside it "
                                                  "(GS_TWO_PHASE_STEP=0, untimed pass of this run)"}
                        blend[kname] = v
            if dom in blend:
                roofline.update(blend[dom])
                roofline["hbm_view"] = hbm
            if prof_alone is not None and "step_uninstanced" in stages:
                # the kernel that runs BESIDE render_bwd in the two-phase step, priced by the bytes the counters saw
                co = {"kernel": "step_uninstanced", "ms_per_launch": stages["step_uninstanced"]["ms_per_launch"], "bound": "hbm",
                      "what": "Adam update (zero gradient) + view statistics of the Gaussians without instances: streams "
                              "their parameters and both moments on a side stream while render_bwd issues vector "
                              "instructions; gs_backward_step's per-Gaussian kernel waits for both"}
                try:
                    t = json.load(open(tf)) if args.config == "c3" else {}  # (the counters were collected on the C3 workload)
                    why = counters_match(t, P, R_last) if t else "no counters for this configuration"
                    if why is not None:
                        co["traffic"], co["traffic_dropped"] = None, why
                    elif t.get("step_uninstanced"):
                        co["traffic"] = t["step_uninstanced"]
                        co["achieved"] = t["step_uninstanced"] / 1e9 / (co["ms_per_launch"] / 1e3)
                        co["peak"], co["unit"] = HBM_PEAK_GBS, "GB/s"
                        co["frac"] = co["achieved"] / HBM_PEAK_GBS
                except Exception:
                    pass
                roofline["co_running"] = co
            roofline["blend_kernels_valu"] = blend or None
        out = {
            "metric": "train-step views/s (fwd+bwd) @1M Gaussians 1080p" if args.config == "c3"
                      else "train-step views/s (fwd+bwd) [%s]" % args.config,
            "value": value, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "gaussians": P, "image": "%dx%d" % (W, H),
                       "scene": SCENES[main_scene]["what"], "sh_degree": SCENES[main_scene]["sh_degree"], "cameras_per_step": world,
                       "dormant_blocks": dormant_info,
                       "row_order": ("Morton order of the centres (GaussianModelLite.spatial_order, the model's default from 100 000 "
                                     "Gaussians on: a permutation of the generated scene; GS_BENCH_SPATIAL_ORDER=0 = as generated)"
                                     if getattr(tr.model, "spatial_order", False) else "as generated"),
                       # which instance lists the timed step renders from, and the same step on the reference's own lists
                       "lists": ("depth-limited (exact, verified per view: csrc/gs_tilecull.h, tests/test_gpu_fullsize.py::"
                                 "test_c3_benched_step_is_the_unlimited_run)" if depth_limit else
                                 "culled (exact: no pixel of a dropped pair reaches alpha >= 1/255)"),
                       "reference_lists_views_per_s": None if ref_lists is None else ref_lists["views_per_s"],
                       "reference_lists_ms_per_step": None if ref_lists is None else ref_lists["ms_per_step"],
                       "num_rendered_reference_lists": None if ref_lists is None else ref_lists["num_rendered_last_view"],
                       "num_rendered_last_view": R_last, "loss": "L1+SSIM" + ("+DWT2" if dwt else "") +
                       ("+patchDWT" if patch else ""),
                       "optimizer": "Adam eps 1e-15, " + (
                           "fused HIP kernel over this rank's shard of the flat buffer, gated by the reduced validity flag "
                           "(the backward kernel writes gradients, statistic increments and flag into the exchange buffer)"
                           if world > 1 else ("inside the backward's per-Gaussian kernel (gs_backward_step)"
                                              if "preprocess_bwd_step" in stages else "fused HIP kernel over the flat buffer")),
                       "parallelism": ("camera-sharded dp%d, visibility-sparse exchange: mask all-reduce, the union's gradient rows "
                                       "packed and all-reduced, dense gated Adam on every replica" % world)
                       if getattr(tr, "sparse_exchange", False) and world > 1 else
                       ("camera-sharded dp%d, reduce-scatter of 59 f32/Gaussian + Adam on 1/N of the rows + "
                        "all-gather of the parameters (GS_SHARDED_ADAM=1)" % world)
                       if getattr(tr, "sharded_optimizer", False) and world > 1 else
                       "camera-sharded dp%d, one all-reduce of 61 f32/Gaussian (59 gradients + 2 statistic increments)" % world},
            "roofline": roofline,
            "reference_lists": ref_lists,
            "options": options,
            "drop_in_api": drop_in,
            "drop_in_api_ms_per_step": None if drop_in is None else drop_in["drop_in_api_ms_per_step"],
            "other_scenes": other or None,
            "sustained": sustained,
            "data_parallel": dp_info,
            "step_times_ms": {"timed (config.lists)": dt / args.steps * 1e3,
                              "reference lists": None if ref_lists is None else ref_lists["ms_per_step"],
                              "init-like, sh_degree 0": other.get("init_like", {}).get("ms_per_step"),
                              "unsaturated background": other.get("ball_in_shell", {}).get("ms_per_step")},
            "stages": stages,
            "stages_one_launch_step": None if prof_alone is None else {
                "ms_per_launch": prof_alone,
                "preprocess_bwd_step_GBps": (None if "preprocess_bwd_step" not in prof_alone else
                                             stage_bytes(P, R_last, N)["preprocess_bwd_step"] / 1e9 /
                                             (prof_alone["preprocess_bwd_step"] / 1e3)),
                "what": "the same step with the per-Gaussian stage as ONE launch after the blend (GS_TWO_PHASE_STEP=0): in the "
                        "timed step `step_uninstanced` (the Adam update of the Gaussians without instances, an HBM stream) "
                        "runs on a side stream beside `render_bwd` (bound by vector issue) - both are slower than alone, "
                        "the pair is faster than one after the other - and `preprocess_bwd_step` only steps the Gaussians "
                        "with instances"},
            "stages_note": ("HIP events around every kernel group in an untimed EAGER pass of the same step right after the timed "
                            "region (each event pair drains the pipeline for ~10 us); the timed region replays the step "
                            "from a hipGraph (%d replays, %d eager fall-backs, %d captures), so no event sits inside it; "
                            "the timed step includes the optimizer" % (graphed.replays, graphed.eager_steps, graphed.captures))
                           if graphed is not None else
                           ("HIP events; %s measured inside the timed region (on every 4th step of it), the other stages in a "
                            "separate untimed pass of the same step right after it (each event pair drains the pipeline for ~10 us); "
                            "the timed step includes the optimizer" % dom_stage),
            "launch": "hipGraph replay of the captured step" if graphed is not None else "eager",
            "depth_limit": ({"mode": "deferred verdict", "limited_views_in_timed_region": dl1["used"] - dl0["used"],
                             "fallbacks_in_timed_region": dl1["failed"] - dl0["failed"],
                             "what": "on a camera's later visits, (tile, Gaussian) pairs behind the depth at which the tile's "
                                     "blend stopped last time are not emitted; the forward verifies the cut lists, a failed "
                                     "view is stepped again with full lists (csrc/gs_tilecull.h, tests/test_gpu_depth_limit.py); "
                                     "GS_BENCH_DEPTH_LIMIT=0 = full lists"} if depth_limit else None),
            "launch_trial": graph_choice,
        }
        if world == 1 and not args.no_cpu_baseline and args.config not in NIR_CONFIGS:
            ci = tr.camera_index(k - 1)
            out["cpu_baseline"] = cpu_baseline(args.config, scene, cams[ci], gts[ci], log)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
