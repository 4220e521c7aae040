"""Developer tool / report generator: is the HIP-vs-oracle PSNR gap through densification (0.19 dB held-out / 0.44 dB train in
round 4, against the 0.05 dB bar of SURVEY 8d) threshold chaos or an arithmetic difference?  The decisive experiment
(tests/psnr_protocol.py): the same 300-iteration densifying protocol
  1. on the HIP backend, free (its clone / split / prune masks recorded),
  2. on the CPU oracle, free (the round-4 comparison: where do the two part, and how close to the thresholds are the Gaussians
     they decide differently?),
  3. on the CPU oracle with HIP's decisions replayed (same discrete trajectory: what is left is arithmetic),
  4. on the HIP backend with the free oracle's decisions replayed (the converse),
  5. on the CPU oracle, free, a second time with another OpenMP thread count (the reference arithmetic against ITSELF: its
     double-precision sums are added in a thread-dependent order, a last-bit perturbation - the yardstick for chaos).
Writes gpurun_out/<tag>_psnr_replay.json.
    python tests/tools/psnr_replay.py [iterations] [tag]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sparse-view-3dgs-pack_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import oracle_lib  # noqa: E402
import psnr_protocol as pp  # noqa: E402
from gsplat_amd import hip_backend  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
TAG = sys.argv[2] if len(sys.argv) > 2 else "r05"
hip, orc = hip_backend(), oracle_lib.get()
cuda, cpu = torch.device("cuda"), torch.device("cpu")


def log(s):
    print(s, flush=True)


def hip_run(tag, replay=None):
    return pp.run(cuda, dgr.GaussianRasterizer, dgr.GaussianRasterizationSettings, hip.api, ITERS, replay=replay, tag=tag, log=log)


def oracle_run(tag, replay=None):
    return pp.run(cpu, orc.Rasterizer, orc.Settings, orc.api, ITERS, replay=replay, tag=tag, log=log)


def per_densification(run):
    """replayed run: what the run's own statistics would have decided, against what it was told to do"""
    out = {}
    for it, d in sorted(run["decisions"].items()):
        out[it] = dict(disagree=d["disagree"], straddle=pp.straddlers(d["own"], d["replay"]), gaussians=int(d["own"]["clone"].numel()),
                       clones=int(d["replay"]["clone"].sum()), splits=int(d["replay"]["split"].sum()),
                       pruned=int(d["replay"]["prune"].sum()))
    return out


t0 = time.perf_counter()
h = hip_run("hip free      ")
o = oracle_run("oracle free   ")
t1 = time.perf_counter()
o_rep = oracle_run("oracle<-hip   ", replay=h["decisions"])
h_rep = hip_run("hip<-oracle   ", replay=o["decisions"])
threads = torch.get_num_threads()
os.environ["OMP_NUM_THREADS"] = str(max(1, threads // 2))
try:
    import ctypes
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(max(1, (os.cpu_count() or 2) // 2))
except OSError:
    pass
o2 = oracle_run("oracle free #2")
it_div, strad = pp.first_divergence(h, o)
it_div2, strad2 = pp.first_divergence(o, o2)
rows = lambda r: r["rows"]
rep = dict(protocol="SURVEY 8d PSNR parity through densification, config-1 size (10k Gaussians, 400x400, 3 train / 3 held-out "
                    "views), L1+SSIM+DWT2+patchDWT, Adam, densification every 40 it from 60, opacity reset every 150",
           iterations=ITERS,
           free=dict(hip=rows(h), oracle=rows(o), gap_test_train=pp.psnr_gap(h, o), first_diverging_densification=it_div,
                     straddlers_there=strad),
           oracle_with_hip_decisions=dict(rows=rows(o_rep), gap_test_train_vs_hip=pp.psnr_gap(h, o_rep),
                                          per_densification=per_densification(o_rep)),
           hip_with_oracle_decisions=dict(rows=rows(h_rep), gap_test_train_vs_oracle=pp.psnr_gap(o, h_rep),
                                          per_densification=per_densification(h_rep)),
           oracle_vs_oracle_other_thread_count=dict(rows=rows(o2), gap_test_train=pp.psnr_gap(o, o2),
                                                    first_diverging_densification=it_div2, straddlers_there=strad2,
                                                    bit_identical=bool(torch.equal(o["flat"], o2["flat"]))
                                                    if o["flat"].shape == o2["flat"].shape else False),
           seconds=dict(hip_and_oracle_free=t1 - t0, total=time.perf_counter() - t0))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
name = "%s_psnr_replay.json" % TAG
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)
print("free: HIP vs oracle max |dPSNR| test %.4f / train %.4f dB (first diverging densification: %s, %s)" %
      (rep["free"]["gap_test_train"] + (it_div, strad)))
print("oracle with HIP's decisions vs HIP: test %.4f / train %.4f dB" % rep["oracle_with_hip_decisions"]["gap_test_train_vs_hip"])
print("HIP with the oracle's decisions vs oracle: test %.4f / train %.4f dB" % rep["hip_with_oracle_decisions"]["gap_test_train_vs_oracle"])
print("oracle vs oracle (other thread count): test %.4f / train %.4f dB, first divergence %s" %
      (rep["oracle_vs_oracle_other_thread_count"]["gap_test_train"] + (it_div2,)), name)
