"""The data-parallel path (cameras sharded over ranks, one all-reduce of the flat 59-float/Gaussian
gradient buffer + densification statistics) on CPU: world_size 2, gloo backend, oracle rasterizer.
Checks that the all-reduced gradient equals the single-process sum of the per-camera gradients and
that replicas stay bit-identical after the optimizer step."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir, sharded=False, sparse=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from test_trainer_cpu import make_trainer
    orc = oracle_lib.get()
    tr = make_trainer(orc, P=400, W=96, H=64, world_size=world, rank=rank)
    tr.sharded_optimizer, tr.sparse_exchange = sharded, sparse
    for k in range(2):
        tr.step(k)
    m = tr.model
    torch.save(dict(flat=m.flat.clone(), grad=m.flat_grad.clone(), m1=m.optimizer.exp_avg.clone(),
                    m2=m.optimizer.exp_avg_sq.clone(), accum=m.xyz_gradient_accum.clone(),
                    denom=m.denom.clone(), maxr=m.max_radii2D.clone(), cams=[tr.camera_index(k) for k in range(2)],
                    exchange=tr.last_exchange),
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_allreduce_matches_single_process_sum(oracle):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    # replicas identical after two steps (same all-reduced gradients, same Adam)
    for k in ("flat", "grad", "accum", "denom", "maxr"):
        assert torch.equal(r0[k], r1[k]), k
    assert r0["cams"] == [0, 2] and r1["cams"] == [1, 3]   # camera sharding: k*world + rank

    # single process: the same two steps with the gradients of both cameras summed by hand
    from test_trainer_cpu import make_trainer
    single = [make_trainer(oracle, P=400, W=96, H=64, world_size=1, rank=0) for _ in range(world)]
    for tr in single:
        tr.optimizer_step = False
    for k in range(2):
        for r, tr in enumerate(single):
            ci = (k * world + r) % len(tr.cameras)
            tr.camera_index = (lambda c: (lambda _k: c))(ci)
            tr.step(k)
        total = single[0].model.flat_grad + single[1].model.flat_grad
        for tr in single:
            tr.model.flat_grad.copy_(total)
            tr.model.optimizer.step()
    # statistics: every replica holds the sums over ALL cameras of ALL steps (each single-process trainer
    # accumulated only its own cameras)
    acc = single[0].model.xyz_gradient_accum + single[1].model.xyz_gradient_accum
    den = single[0].model.denom + single[1].model.denom
    mxr = torch.maximum(single[0].model.max_radii2D, single[1].model.max_radii2D)
    # gloo sums two fp32 buffers: a + b is exact and commutative, so the results agree bit for bit
    assert torch.equal(single[0].model.flat_grad, r0["grad"])
    assert torch.equal(single[0].model.flat, r0["flat"])
    assert torch.equal(den, r0["denom"]) and torch.equal(mxr, r0["maxr"])
    assert float(den.max()) >= 2.0  # a Gaussian seen by several cameras counts once per camera, not 2^steps
    assert torch.allclose(acc, r0["accum"], rtol=1e-6, atol=1e-12)


@pytest.mark.timeout(600)
def test_sharded_optimizer_equals_the_all_reduce_path_bit_for_bit():
    """reduce-scatter -> Adam on this rank's 1/N of the flat rows -> all-gather of the parameters (DESIGN.md 5,
    Trainer._exchange_and_step_sharded) against all-reduce + the full Adam pass on every replica: after two steps on two
    ranks the parameters and the statistics are the same bits on every rank of both runs; each rank's moments are
    those of the all-reduce run on its own shard (and untouched elsewhere)."""
    world = 2
    runs = {}
    for sharded in (False, True):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker, args=(world, _free_port(), d, sharded), nprocs=world, join=True)
            runs[sharded] = [torch.load(os.path.join(d, "rank%d.pt" % r)) for r in range(world)]
    ref = runs[False][0]
    n = ref["flat"].numel()
    from gsplat_amd.trainer import SHARD_UNIT
    S = ((n + SHARD_UNIT - 1) // SHARD_UNIT * SHARD_UNIT) // world
    for r in range(world):
        got = runs[True][r]
        for k in ("flat", "accum", "denom", "maxr"):
            assert torch.equal(got[k], ref[k]), (r, k)
        lo, hi = r * S, min((r + 1) * S, n)
        for k in ("m1", "m2"):
            assert torch.equal(got[k][lo:hi], ref[k][lo:hi]), (r, k)
            other = torch.cat((got[k][:lo], got[k][hi:]))
            assert float(other.abs().max()) == 0.0  # this rank never touched the other shards' moments
        assert torch.equal(got["grad"][lo:hi], ref["grad"][lo:hi])
    assert 0 < S < n  # both ranks own a non-empty shard


def _worker_schedule(rank, world, port, outdir, sharded=False, sparse=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from gsplat_amd.trainer import TrainOptions
    from test_trainer_cpu import make_trainer
    orc = oracle_lib.get()
    tr = make_trainer(orc, P=300, W=96, H=64, world_size=world, rank=rank, dwt=False)
    tr.sharded_optimizer, tr.sparse_exchange = sharded, sparse
    opt = TrainOptions(iterations=20, densify_from_iter=2, densification_interval=3, opacity_reset_interval=5,
                       densify_until_iter=12, cameras_extent=4.4, densify_grad_threshold=1e-7, seed=1)
    cams, sizes = [], []
    for it in range(1, 9):
        out = tr.train_iteration(it, opt)
        cams.append(out["camera"])
        sizes.append(out["P"])
    tr.gather_optimizer_state()
    m = tr.model
    torch.save(dict(flat=m.flat.clone(), m1=m.optimizer.exp_avg.clone(), m2=m.optimizer.exp_avg_sq.clone(), cams=cams,
                    sizes=sizes), os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_optimizer_through_densification_equals_the_all_reduce_path():
    """8 iterations of the schedule (densify / prune / opacity reset: the shard boundaries move with P, so the moments
    are all-gathered before every re-layout) on two ranks, sharded against all-reduce: same sizes, parameters and
    moments, bit for bit."""
    out = {}
    for sharded in (False, True):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker_schedule, args=(2, _free_port(), d, sharded), nprocs=2, join=True)
            out[sharded] = [torch.load(os.path.join(d, "rank%d.pt" % r)) for r in range(2)]
    a = out[False][0]
    assert a["sizes"][-1] != a["sizes"][0]
    for r in range(2):
        b = out[True][r]
        assert b["sizes"] == a["sizes"]
        for k in ("flat", "m1", "m2"):
            assert torch.equal(a[k], b[k]), (r, k)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_schedule_with_densification_keeps_replicas_identical(world):
    """Densify / prune / opacity reset run from all-reduced statistics and a shared seed: after 8 iterations of the
    schedule all replicas hold the same number of Gaussians, bit-identical parameters and Adam moments, and each
    global step consumed `world` different cameras of the shared draw-without-replacement stack."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_schedule, args=(world, _free_port(), d), nprocs=world, join=True)
        rs = [torch.load(os.path.join(d, "rank%d.pt" % r)) for r in range(world)]
    r0 = rs[0]
    assert r0["sizes"][-1] != r0["sizes"][0]
    for r in rs[1:]:
        assert r["sizes"] == r0["sizes"]
        for k in ("flat", "m1", "m2"):
            assert torch.equal(r0[k], r[k]), k
    per_step = list(zip(*[r["cams"] for r in rs]))  # cameras of ranks 0..world-1 at each global step
    for cams in per_step:
        assert len(set(cams)) == world
    flat = [c for cams in per_step for c in cams]
    for k in range(0, len(flat), 4):  # 4 cameras in the scene: every 4 draws are a permutation of them
        assert sorted(flat[k:k + 4]) == [0, 1, 2, 3]


@pytest.mark.timeout(600)
def test_sparse_exchange_equals_the_dense_all_reduce_bit_for_bit():
    """The visibility-sparse exchange (Trainer._exchange_and_step_sparse: union of the ranks' row masks, pack, all-reduce,
    scatter, dense Adam) against the dense all-reduce on two ranks: gradients, parameters, both moments and the statistics
    are the same bits on every rank (a + b commutes; rows outside the union are 0 + 0), and fewer bytes moved."""
    world = 2
    runs = {}
    for sparse in (False, True):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker, args=(world, _free_port(), d, False, sparse), nprocs=world, join=True)
            runs[sparse] = [torch.load(os.path.join(d, "rank%d.pt" % r)) for r in range(world)]
    ref = runs[False][0]
    for r in range(world):
        got = runs[True][r]
        for k in ("flat", "grad", "m1", "m2", "accum", "denom", "maxr"):
            assert torch.equal(got[k], ref[k]), (r, k)
        ex = got["exchange"]
        # (this small scene lies inside every camera's frustum: the union is all of it; tests/test_gpu_dp_fused.py runs the
        #  depth-limited GPU form, where a view reaches a fifth of the Gaussians)
        assert 0 < ex["union_rows"] <= ex["rows"] and ex["sparse_bytes"] <= ex["dense_bytes"] + ex["rows"]
    assert runs[False][0]["exchange"] is None


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_sparse_exchange_through_densification(world):
    """8 iterations of the schedule (densify / prune / opacity reset: P changes, the masks and the packed buffer follow) with
    the sparse exchange: replicas bit-identical; against the dense all-reduce run - the same bits on two ranks, and on
    four (where the library adds the four contributions in an order of its own choosing in either form) the same sizes
    and parameters to rounding."""
    out = {}
    for sparse in (False, True):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker_schedule, args=(world, _free_port(), d, False, sparse), nprocs=world, join=True)
            out[sparse] = [torch.load(os.path.join(d, "rank%d.pt" % r)) for r in range(world)]
    a = out[False][0]
    assert a["sizes"][-1] != a["sizes"][0]
    for r in range(world):
        b = out[True][r]
        assert b["sizes"] == out[True][0]["sizes"]
        for k in ("flat", "m1", "m2"):
            assert torch.equal(out[True][0][k], b[k]), (r, k)      # replicas
    b = out[True][0]
    if world == 2:
        assert b["sizes"] == a["sizes"]
        for k in ("flat", "m1", "m2"):
            assert torch.equal(a[k], b[k]), k
    else:
        assert b["sizes"][:3] == a["sizes"][:3]
        if b["sizes"] == a["sizes"]:
            d = (a["flat"] - b["flat"]).double()
            assert float(d.pow(2).mean().sqrt()) <= 1e-3 * float(a["flat"].double().pow(2).mean().sqrt())


def _worker_partial_union(rank, world, port, outdir, sparse, split):
    """As _worker, on a scene a third of whose Gaussians no camera sees (100 units above the orbit): the union of the ranks'
    row masks is a proper subset, so the two parts of the sparse exchange's optimizer both have rows to step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from test_trainer_cpu import make_trainer
    orc = oracle_lib.get()
    tr = make_trainer(orc, P=400, W=96, H=64, world_size=world, rank=rank)
    tr.sharded_optimizer, tr.sparse_exchange, tr.SPLIT_SPARSE_ADAM = False, sparse, split
    with torch.no_grad():
        tr.model.params["xyz"][::3, 2] += 100.0
        # every row carries momentum from "earlier steps": the zero-gradient update of an unseen row then changes its bits
        g = torch.Generator().manual_seed(7)
        opt = tr.model.optimizer
        opt.exp_avg.copy_(torch.randn(opt.exp_avg.shape, generator=g) * 1e-3)
        opt.exp_avg_sq.copy_(torch.rand(opt.exp_avg_sq.shape, generator=g) * 1e-6 + 1e-9)
        opt.t = 5
        for name in opt.seg_steps:
            opt.seg_steps[name] = 5
    xyz_start = tr.model.params["xyz"].detach().clone()
    unions = []
    for k in range(3):
        tr.step(k)
        unions.append(None if tr.last_exchange is None else tr.last_exchange["union_rows"])
    m = tr.model
    torch.save(dict(flat=m.flat.clone(), grad=m.flat_grad.clone(), m1=m.optimizer.exp_avg.clone(),
                    m2=m.optimizer.exp_avg_sq.clone(), accum=m.xyz_gradient_accum.clone(), denom=m.denom.clone(),
                    maxr=m.max_radii2D.clone(), unions=unions, t=m.optimizer.t, seg_steps=dict(m.optimizer.seg_steps),
                    xyz_start=xyz_start, xyz_now=m.params["xyz"].detach().clone(), union_mask=m.union_mask.clone()),
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sparse_exchange_steps_the_rows_outside_the_union_under_the_collective():
    """Trainer._exchange_and_step_sparse applies Adam in two parts - the rows outside the union while the union's rows are on the
    links, the union's rows when their sums are back - and must leave the bits of ONE dense pass: against the dense all-reduce
    form and against the sparse exchange with a single pass (SPLIT_SPARSE_ADAM = False), with a union that is a proper subset
    (the rows outside it still move: their moments decay, their parameters follow the old momentum)."""
    world = 2
    runs = {}
    for key, (sparse, split) in dict(dense=(False, True), one_pass=(True, False), two_parts=(True, True)).items():
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker_partial_union, args=(world, _free_port(), d, sparse, split), nprocs=world, join=True)
            runs[key] = [torch.load(os.path.join(d, "rank%d.pt" % r)) for r in range(world)]
    ref = runs["dense"][0]
    assert ref["t"] == 5 + 3
    for key in ("one_pass", "two_parts"):
        for r in range(world):
            got = runs[key][r]
            assert all(0 < u < 400 for u in got["unions"]), got["unions"]
            assert got["t"] == ref["t"] and got["seg_steps"] == ref["seg_steps"]
            for k in ("flat", "grad", "m1", "m2", "accum", "denom", "maxr"):
                assert torch.equal(got[k], ref[k]), (key, r, k)
    # the rows nobody sees were stepped too (their momentum moved them), by the part that runs under the collective
    got = runs["two_parts"][0]
    unseen = got["union_mask"] == 0
    assert int(unseen.sum()) >= 400 // 3
    assert bool((got["xyz_now"][unseen] != got["xyz_start"][unseen]).any(dim=1).all())


def _worker_eight(rank, world, port, outdir, form):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from test_trainer_cpu import make_trainer
    tr = make_trainer(oracle_lib.get(), P=203, W=64, H=48, world_size=world, rank=rank, dwt=False)   # 203: no multiple of 8
    tr.sharded_optimizer, tr.sparse_exchange = form == "sharded", form == "sparse"
    with torch.no_grad():
        tr.model.params["xyz"][::4, 2] += 100.0          # a quarter of the rows outside every view
    for k in range(2):
        tr.step(k)
    tr.gather_optimizer_state()
    m = tr.model
    torch.save(dict(flat=m.flat.clone(), m1=m.optimizer.exp_avg.clone(), m2=m.optimizer.exp_avg_sq.clone(),
                    accum=m.xyz_gradient_accum.clone(), denom=m.denom.clone(), maxr=m.max_radii2D.clone(),
                    cams=[tr.camera_index(k) for k in range(2)]), os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_eight_ranks_every_exchange_form():
    """The driver's scaling run goes to N = 8: eight gloo ranks (one thread each), a row count that is no multiple of eight, a
    quarter of the rows seen by nobody.  Replicas bit-identical in every form; sharded and sparse against the all-reduce form:
    the same parameters, moments and statistics to rounding (eight addends: the library chooses the order in each form)."""
    world = 8
    runs = {}
    for form in ("allreduce", "sharded", "sparse"):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker_eight, args=(world, _free_port(), d, form), nprocs=world, join=True)
            runs[form] = [torch.load(os.path.join(d, "rank%d.pt" % r)) for r in range(world)]
        r0 = runs[form][0]
        for r in range(1, world):
            for k in ("flat", "m1", "m2", "accum", "denom", "maxr"):
                assert torch.equal(r0[k], runs[form][r][k]), (form, r, k)
        assert sorted(c for rr in runs[form] for c in rr["cams"][:1]) == sorted(k % 4 for k in range(8))   # k * 8 + r over 4 cameras
    ref = runs["allreduce"][0]
    for form in ("sharded", "sparse"):
        got = runs[form][0]
        assert torch.equal(got["denom"], ref["denom"]) and torch.equal(got["maxr"], ref["maxr"])
        for k in ("flat", "m1", "m2", "accum"):
            d = (got[k] - ref[k]).double()
            assert float(d.abs().max()) <= 1e-5 * max(float(ref[k].abs().max()), 1e-30), (form, k)
