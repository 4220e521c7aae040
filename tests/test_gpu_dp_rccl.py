"""The collectives of the data-parallel step THROUGH RCCL.  The test box has one GPU and RCCL refuses two ranks on one
device, so tests/test_gpu_dp_fused.py runs its two ranks over gloo - every kernel and the whole control flow, but not the
library the multi-GPU run uses.  Here ONE process forms a one-rank `nccl` group (= RCCL on ROCm) and a Trainer that is TOLD it
is rank 0 of 2 runs the fused data-parallel step: every all-reduce the step issues - uint8 MAX of the row masks on the side
stream, float SUM of slices of the exchange buffer (blocking and async_op), float MAX of max_radii2D, the packed union rows
with the optimizer's first part enqueued under it - goes through RCCL's API with the dtypes, reduction ops, views and streams
of the real run; a one-rank sum is the identity, so the result must be, bit for bit, what the same lying Trainer gets over a
one-rank gloo group."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_gpu_dp_fused import _free_port, _make

pytestmark = pytest.mark.gpu


def _worker(rank, backend, port, outdir, form):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    if backend == "nccl":   # exactly bench.py's call (eager communicator on the rank's device)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=0, world_size=1)
    tr, hip = _make(0, 2, fused=True)        # told: rank 0 of 2 (cameras 0, 2, 0, 2 ...); the group has one member
    tr.sharded_optimizer, tr.sparse_exchange = False, form == "sparse"
    hip.tile_cull = True
    tr.depth_limit = "deferred"
    assert tr._fused_dp_ok(hip, True)
    for k in range(6):
        tr.step(k)
    tr.sync()
    torch.cuda.synchronize()
    m = tr.model
    torch.save(dict(flat=m.flat.detach().cpu(), m1=m.optimizer.exp_avg.cpu(), m2=m.optimizer.exp_avg_sq.cpu(),
                    accum=m.xyz_gradient_accum.cpu(), denom=m.denom.cpu(), maxr=m.max_radii2D.cpu(), t=m.optimizer.t,
                    exchange=tr.last_exchange, used=hip.depth_limit_stats["used"]),
               os.path.join(outdir, "%s_%s.pt" % (backend, form)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("form", ["allreduce", "sparse"])
def test_the_data_parallel_step_issues_its_collectives_through_rccl(hip, form):
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for backend in ("gloo", "nccl"):
            mp.spawn(_worker, args=(backend, _free_port(), d, form), nprocs=1, join=True)
            out[backend] = torch.load(os.path.join(d, "%s_%s.pt" % (backend, form)))
    a, b = out["gloo"], out["nccl"]
    assert a["t"] == b["t"] == 6 and b["used"] > 0
    for k in ("flat", "m1", "m2", "accum", "denom", "maxr"):
        assert torch.equal(a[k], b[k]), k
    if form == "sparse":
        ex = b["exchange"]
        assert 0 < ex["union_rows"] < ex["rows"] and ex["sparse_bytes"] < ex["dense_bytes"]
