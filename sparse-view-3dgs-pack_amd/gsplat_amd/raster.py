"""Host glue between torch tensors and the C ABI (include/gsplat.h).

``RasterBackend`` re-creates the three pybind entry points of the reference extension module
(`rasterize_gaussians`, `rasterize_gaussians_backward`, `mark_visible`:
diff-gaussian-rasterization/ext.cpp:15-19, rasterize_points.cu:35-244) with identical positional
arguments and return tuples, on top of the two-phase C ABI.  Tensors stay torch-owned; only raw
addresses and the current HIP stream cross the boundary.
"""
import ctypes as C
import os

import torch

from .capi import GsGaussians, GsGrads, GsScratch, GsView

NUM_CHANNELS = 3


def _ptr(t):
    """Explicit NULL for 'absent' (the reference relied on data_ptr() of an empty tensor)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _prep(t, device):
    """contiguous fp32 on `device`; empty stays empty (rasterize_points.cu:101-120)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    if t.device != device:
        t = t.to(device)
    return t.contiguous()


class RasterBackend:
    """Binds one implementation of the C ABI (`api`) to one torch device type."""

    def __init__(self, api, device_type="cuda"):
        self.api = api
        self.device_type = device_type
        self._pinned = None
        # optimistic binning capacity (instances): skips the forward's host sync when the previous
        # call's num_rendered is a good predictor; see rasterize_gaussians().
        self._capacity_hint = 0
        self._capacity_hint_limited = 0
        self.optimistic = os.environ.get("GS_SYNC_FORWARD", "0") != "1"
        # instance lists: True = drop (tile, Gaussian) pairs no pixel of which can reach alpha >= 1/255
        # (csrc/gs_tilecull.h; same images / gradients, ~2.6x fewer instances); GS_TILE_CULL=0 = the
        # reference's bounding-square lists, bit-identical point_list / ranges / num_rendered
        self.tile_cull = os.environ.get("GS_TILE_CULL", "1") != "0"
        # how the culled lists are built: "region" = region binning (csrc/gs_regionbin.hip, GsView.tile_cull = 2: two launches),
        # "lsd" = depth sort + emission + partition by tile (csrc/gs_binning.hip, 23 launches).  Same lists tile by tile.
        # "auto" (default): region binning, except for forwards WITHOUT depth limits whose lists are known to be long (more
        # than REGION_AUTO_MAX instances per region at the capacity the previous views needed: ~5 000 Gaussians per region) -
        # a region's bitonic sort grows as n log^2 n and its bucket atomics as n, so beyond that the LSD path is as fast or
        # faster (C3 full culled lists, 9.55 M instances: preprocess + binning 0.44 ms with regions, 0.42 ms LSD; depth-limited,
        # 1.86 M: 0.20 vs 0.30)
        self.binning = os.environ.get("GS_BINNING", "auto")
        self.last_deferred_num_rendered = None  # instance count of the last deferred forward whose verdict was collected
        self._region_off = set()   # (P, W, H, limited lists?) whose regions hold more Gaussians than one workgroup sorts: LSD path
        self._cap_memo = {}
        self._scratch_memo = {}
        self._cap_by_buffer = {}
        self._cam_cache = {}
        # one-shot identity of the camera of the NEXT forward (GaussianRasterizer.camera_key); None = hash the view matrix
        self.camera_key = None
        # one-shot, set with camera_key by GaussianRasterizer: the caller named the camera and did not decline depth limits
        # (GaussianRasterizer.camera_limits) - the forward may use this camera's verified limits as with GS_DEPTH_LIMIT=1
        self.camera_key_limits = False
        self._region_key = None    # (P, W, H, limited) of the forward being served (see _region_off); None in the backward
        self._vm_ids = {}
        self.camera_cache_stats = dict(hits=0, misses=0, hashed=0)
        self.order_hint_on = True   # (False: image order - outputs do not depend on it, see GsScratch.tile_order_hint)
        # depth-limited emission (GsScratch.tile_depth_limit): on the second and later visits of a camera, (tile, Gaussian)
        # pairs behind the depth at which that tile's blend stopped last time are not emitted.  Checked, not assumed: the
        # forward flags lists that were cut too short and the view is rendered again without limits - which the host
        # can only know once the blend has finished.  Waiting for that inside the forward costs more than the shorter sort
        # saves (C3: 1.43 -> 1.60 ms/step), so it is OFF unless asked for (GS_DEPTH_LIMIT=1), and the train step asks for
        # the deferred form instead: `depth_limit_request = "defer"` (one-shot, next forward) leaves the verdict in
        # `self.deferred` for the caller to collect later - gsplat_amd.trainer does, its step kernel being a no-op on the
        # device when the flag is set.  Only with tile_cull.
        self.depth_limit_on = os.environ.get("GS_DEPTH_LIMIT", "0") == "1"
        self.depth_limit_request = None
        self.deferred = None
        self._status_ring = {}
        self.depth_limit_stats = dict(used=0, failed=0)
        self._pinned_by_device = {}
        # one-shot output arena for the next backward: {"means3D": [P,3], "sh": [P,M,3]} fp32 tensors to write
        # dL_dmeans3D / dL_dsh INTO (e.g. views of a flat gradient buffer) instead of fresh allocations
        self.grad_arena = None
        # capture-safe forward (hipGraph): a fixed binning capacity (instances) instead of the predicted one, no host
        # wait, no host read.  The forward then returns the CAPACITY in place of num_rendered (the kernels read the real
        # count from the device header; the backward only uses the value as an upper bound) and never re-runs: the
        # caller checks `last_num_rendered()` against the capacity after the stream has drained.
        self.static_capacity = None
        self.static_step_tag = None   # capture-safe mode: device word the captured gs_forward_status copies out with the status
        # one-shot request for the next backward: a gsplat_amd.capi.GsStepState (+ the tensors it points into, kept alive
        # by the caller) - run gs_backward_step (backward + activation backward + view statistics + Adam in the same
        # per-Gaussian kernel) instead of gs_backward; the backward then returns no gradients at all
        self.fused_step = None
        self._side_streams = {}
        self._rows_ws = {}        # (device, bytes) -> [persistent gradient-row workspace of the fused step, rows all zero?]
        self.rows_epoch = 0       # bumped whenever a fused backward found (or may have left) that workspace dirty
        self._early = None        # what launch_uninstanced_early needs of the last forward
        self.two_phase_launches = 0
        self._uninst_done = None
        # one-shot, set together with fused_step by the train step: the opacities / scales / rotations of the next forward
        # (and of its gs_backward_step) are the model's RAW rows, activated inside the kernels
        # (GsGaussians.raw_activations): no activation kernel, no activated copies
        self.raw_activations = False
        self._raw_backward = False
        self._raw_geom = None       # geomBuffer of the last forward on raw rows (whose backward the flag above announces)
        # one-shot, set before a forward / a fused backward on raw rows: the model's _features_rest [P,M-1,3]; `sh` is then
        # _features_dc [P,1,3] (GsGaussians.shs_rest: the kernels read the split rows, no torch.cat)
        self.sh_rest = None
        # parity probes: keep the backward workspace (the per-Gaussian 16-slot float64 gradient rows of the blend backward)
        self.keep_workspace = False
        self.last_workspace = None

    # ------------------------------------------------------------------ helpers
    def _stream(self, device):
        if device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        return None

    def _check_device(self, t):
        if t.device.type != self.device_type:
            raise RuntimeError(
                "gsplat %s backend got a tensor on %s - the rasterizer has no fallback path"
                % (self.device_type, t.device))

    def _view(self, keep, device, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, scale_modifier,
              degree, prefiltered, antialiasing, debug):
        v = GsView()
        v.image_height, v.image_width = int(H), int(W)
        v.tanfovx, v.tanfovy = float(tanfovx), float(tanfovy)
        v.scale_modifier = float(scale_modifier)
        v.sh_degree = int(degree)
        v.prefiltered, v.antialiasing, v.debug = int(bool(prefiltered)), int(bool(antialiasing)), int(bool(debug))
        if self.force_rowwise_entries:   # (test hook: GsView.debug bit 1, csrc/gs_tilebin.hip)
            v.debug |= 2
        v.tile_cull = int(self.tile_cull)
        if self.tile_cull and self.binning in ("region", "auto") and device.type == "cuda" and \
                (self._region_key is None or self._region_key not in self._region_off):
            v.tile_cull = 2
        bg, viewmatrix, projmatrix, campos = (_prep(x, device) for x in (bg, viewmatrix, projmatrix, campos))
        keep += [bg, viewmatrix, projmatrix, campos]
        v.bg, v.viewmatrix, v.projmatrix, v.campos = _ptr(bg), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos)
        return v

    def _gauss(self, keep, device, means3D, sh, colors, opacities, scales, rotations, cov3D, extra=None, raw=False,
               extra_gain=None, sh_rest=None):
        g = GsGaussians()
        g.raw_activations = int(bool(raw))
        extra = _prep(extra, device)
        keep.append(extra)
        g.extra_channel = _ptr(extra)
        if extra_gain is not None:  # raw mode: extra holds the RAW row, activated in the kernels with this device scalar
            extra_gain = _prep(extra_gain, device)
            keep.append(extra_gain)
            g.extra_gain = _ptr(extra_gain)
        means3D, sh, colors, opacities, scales, rotations, cov3D = (
            _prep(x, device) for x in (means3D, sh, colors, opacities, scales, rotations, cov3D))
        keep += [means3D, sh, colors, opacities, scales, rotations, cov3D]
        g.P = int(means3D.shape[0]) if means3D is not None else 0
        g.M = int(sh.shape[1]) if sh is not None else 0
        if sh_rest is not None:
            if sh is None or int(sh.shape[1]) != 1 or sh_rest.shape[0] != sh.shape[0]:
                raise RuntimeError("sh_rest goes with sh = the DC rows [P,1,3]")
            sh_rest = _prep(sh_rest, device)
            keep.append(sh_rest)
            g.M = 1 + int(sh_rest.shape[1])
            g.shs_rest = _ptr(sh_rest)
        g.means3D, g.shs, g.colors_precomp, g.opacities = _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacities)
        g.scales, g.rotations, g.cov3D_precomp = _ptr(scales), _ptr(rotations), _ptr(cov3D)
        return g

    def _update_hint(self, num_rendered, limited=False):
        # 25 % head-room over the largest recent instance count; decays slowly so one huge view does not pin
        # the capacity (and the 16 B/instance buffer) forever.  Quantised to 1/8-octave steps so that consecutive steps
        # ask for the SAME number of bytes: the caching allocator then hands back the same block every step.
        # Depth-limited views keep their own hint: their lists are several times shorter, and the sort's grids, its
        # histogram tables and the row scan are sized by the capacity, not by the count on the device.
        import math
        old = self._capacity_hint_limited if limited else self._capacity_hint
        want = max(int(num_rendered * 1.25) + 4096, int(old * 0.98))
        new = int(2.0 ** (math.ceil(math.log2(want) * 8.0) / 8.0))
        if limited:
            self._capacity_hint_limited = new
        else:
            self._capacity_hint = new

    def _remember_capacity(self, binning, cap):
        """The binning buffer travels through autograd as a bare byte tensor; its capacity (instances) is kept here, keyed
        by the buffer's address and length, so backward need not recover it by search."""
        if len(self._cap_by_buffer) > 256:
            self._cap_by_buffer.clear()
        self._cap_by_buffer[(binning.data_ptr(), binning.numel())] = int(cap)

    def _capacity_of(self, P, W, H, nbytes, at_least):
        """Instances the binning buffer of `nbytes` bytes was sized for (largest cap with bytes(cap) <= nbytes); only
        reached for buffers this backend did not allocate itself (see _remember_capacity)."""
        key = (P, W, H, nbytes)
        cap = self._cap_memo.get(key)
        if cap is None:
            lo, hi = int(at_least), max(int(at_least), 1)
            while self.scratch_bytes(P, W, H, hi)[2] <= nbytes:
                lo, hi = hi, hi * 2
            while lo < hi - 1:
                mid = (lo + hi) // 2
                if self.scratch_bytes(P, W, H, mid)[2] <= nbytes:
                    lo = mid
                else:
                    hi = mid
            cap = lo
            if len(self._cap_memo) > 64:
                self._cap_memo.clear()
            self._cap_memo[key] = cap
        return cap

    def last_num_rendered(self):
        """num_rendered of the most recent forward on the current device / stream as the device reported it (pinned
        host word written by an asynchronous copy: only valid once that forward has completed)."""
        return None if self._pinned is None else int(self._pinned[0])

    def take_deferred(self):
        """The pending verdict of the last forward run with depth_limit_request = "defer", or None.  -> callable that
        waits for that forward and returns True when its limits held (outputs valid) - on False it has already
        invalidated the camera's limits, and the caller must discard everything computed from that forward."""
        d, self.deferred = self.deferred, None
        if d is None:
            return None

        def verdict():
            d["event"].synchronize()
            st = tuple(int(x) for x in d["status"][:4])
            ok = st[1] == 0 and st[2] == 0
            if "regions" in d:
                # region binning learns the instance count only here: keep the capacity hint current, and tell a capacity
                # overflow (render again with more room, the limits were fine) from limits that proved too tight
                if st[3] > self.REGION_MAX_ENTRIES:
                    self._region_off.add(d.get("size"))
                self._update_hint(max(st[0], st[3] * d["regions"]), limited=d.get("limited", False))
                self.last_deferred_num_rendered = st[0]
            if not ok:
                self.depth_limit_stats["failed"] += 1
                if st[2] != 0 or "regions" not in d:
                    d["cache"]["limit_ok"] = False
                if st[2] != 0:
                    self.limits_failed(d["cache"])
            elif d.get("limited", False):
                self.limits_held(d["cache"])
            return ok
        return verdict

    def last_status(self):
        """(num_rendered, overflow, trunc_failed) of the most recent forward that copied its status out
        (gs_forward_status: the capture-safe mode and depth-limited forwards do); valid once it has completed."""
        return None if self._pinned is None else tuple(int(x) for x in self._pinned[:3])

    def _capacity_for(self, binning, P, W, H, R):
        if binning.numel() == 0:
            return R
        cap = self._cap_by_buffer.get((binning.data_ptr(), binning.numel()))
        if cap is not None and cap >= R:
            return cap
        return self._capacity_of(P, W, H, binning.numel(), R)

    def scratch_bytes(self, P, W, H, R):
        # (a pure function of four integers, asked two or three times per forward and once per backward: remembered)
        key = (int(P), int(W), int(H), int(R))
        memo = self._scratch_memo
        got = memo.get(key)
        if got is None:
            out = (C.c_size_t * 3)()
            ws = C.c_size_t(0)
            self.api.call("scratch_bytes", P, W, H, R, out, C.byref(ws))
            if len(memo) > 256:
                memo.clear()
            got = memo[key] = (out[0], out[1], out[2], ws.value)
        return got

    def _camera_identity(self, viewmatrix):
        """Who is this camera?  An explicit key when the caller gave one (GaussianRasterizer.camera_key), else a hash of the
        view matrix's CONTENTS - remembered per tensor object and version, so a loop that keeps one tensor per camera pays
        one 64-byte device-to-host copy per camera, and one that re-creates its camera tensors every step still finds its
        state (at the price of that copy per step; pass a key to avoid it).  An address alone is not an identity: the
        allocator hands a freed block to the next camera."""
        ck, self.camera_key = self.camera_key, None
        if ck is not None:
            return ("key", ck)
        import hashlib
        import weakref
        ent = self._vm_ids.get(id(viewmatrix))
        if ent is not None and ent[0]() is viewmatrix and ent[1] == viewmatrix._version:
            return ent[2]
        self.camera_cache_stats["hashed"] += 1
        ident = ("hash", hashlib.blake2b(viewmatrix.detach().to("cpu", torch.float32).contiguous().numpy().tobytes(),
                                         digest_size=8).hexdigest())
        if len(self._vm_ids) > 1024:
            self._vm_ids = {k: v for k, v in self._vm_ids.items() if v[0]() is not None}
        try:
            self._vm_ids[id(viewmatrix)] = (weakref.ref(viewmatrix), viewmatrix._version, ident)
        except TypeError:
            pass
        return ident

    def _camera_cache(self, device, W, H, viewmatrix):
        """What the previous visit of the SAME camera measured, per tile:
        order - the launch order for the forward blend (GsScratch.tile_order_hint; a hint from another camera is
                worthless: 0.203 ms either way at C3; from the same camera 0.204 -> 0.164 ms, tests/tools/fwd_order_probe.py);
        limit - the depth at which each tile's blend stopped (GsScratch.tile_depth_limit).
        Cameras are told apart by _camera_identity.  A stale or foreign entry costs time, never correctness (the order is
        pure scheduling - any permutation of the tiles renders the same image - and the limits are verified by the
        forward).  Both buffers hold valid contents from the start (the default order, +inf = no limit) and the kernels
        leave them alone when a forward overflowed, so whatever is read from them is safe.
        -> dict(order, order_ok, limit, limit_ok) or None"""
        if device.type != "cuda" or not (self.order_hint_on or self.depth_limit_on or self.depth_limit_request is not None
                                         or self.static_capacity is not None or self.camera_key_limits):
            self.camera_key = None
            return None
        key = (device.index, W, H, self._camera_identity(viewmatrix))
        c = self._cam_cache.get(key)
        if c is None:
            self.camera_cache_stats["misses"] += 1
            if len(self._cam_cache) >= self.CAMERA_CACHE_MAX:
                # forget the oldest camera nobody pinned (a captured hipGraph holds the ADDRESSES of its camera's buffers:
                # GraphedStep pins the entry for as long as the graph lives)
                for k in self._cam_cache:
                    if not self._cam_cache[k].get("pinned"):
                        self._cam_cache.pop(k)
                        break
            T = ((W + 15) // 16) * ((H + 15) // 16)
            per_xcd = (T + 7) // 8
            b = torch.arange(per_xcd * 8, dtype=torch.int64)
            default_order = ((b & 7) * per_xcd + (b >> 3)).to(torch.int32)  # the kernel's own XCD-banded mapping
            c = self._cam_cache[key] = dict(order=default_order.to(device), order_ok=False,
                                            limit=torch.full((int(self.api.raw("tile_depth_limit_floats")(W, H)),), float("inf"),
                                                             dtype=torch.float32, device=device),
                                            limit_ok=False, key=key,
                                            # adaptive slack on the exported bounds (GsScratch.tile_depth_limit_slack)
                                            slack=torch.ones((1,), dtype=torch.float32, device=device), slack_level=0, held=0)
        else:
            self.camera_cache_stats["hits"] += 1
        return c

    CAMERA_CACHE_MAX = 256

    def drop_camera_entries(self, key_prefix):
        """Forget the per-camera state filed under explicit keys that start with `key_prefix` (a trainer that is gone:
        its keys are ("trainer", uid, camera index)); pinned entries too - their graphs died with the trainer."""
        n = len(key_prefix)
        for k in [k for k in self._cam_cache if k[3][0] == "key" and isinstance(k[3][1], tuple) and k[3][1][:n] == key_prefix]:
            self._cam_cache.pop(k, None)

    def camera_entry(self, W, H, viewmatrix=None, camera_key=None, device_index=None):
        """The per-camera state kept for a camera (by explicit key, or by the contents of its view matrix), or None - a
        lookup that creates nothing and counts nothing (tests, tools)."""
        if camera_key is not None:
            ident = ("key", camera_key)
        else:
            saved, self.camera_key = self.camera_key, None
            hashed = self.camera_cache_stats["hashed"]
            ident = self._camera_identity(viewmatrix)
            self.camera_key, self.camera_cache_stats["hashed"] = saved, hashed
        if device_index is None:
            device_index = viewmatrix.device.index if viewmatrix is not None and viewmatrix.is_cuda else torch.cuda.current_device()
        return self._cam_cache.get((device_index, int(W), int(H), ident))

    # Adaptive slack of a camera's depth bounds.  The fixed margin (5 % + 0.02 in view depth) holds while the model moves
    # slowly between two visits of a camera (a run that is converging: 2 repeated views in 2 000 at C3); early in training,
    # or on a target the model cannot fit, tiles keep saturating deeper than their last bound allowed (38 % repeated views
    # on bench.py's unrelated target after a few hundred steps).  So the bounds a camera EXPORTS are multiplied by a factor
    # that rises when its limits fail and falls again after they have held for a while: longer lists instead of repeats.
    SLACK = (1.0, 1.25, 1.6, 2.5)
    SLACK_RELAX_AFTER = 24

    def limits_failed(self, c):
        if c is None:
            return
        c["held"] = 0
        if c["slack_level"] < len(self.SLACK) - 1:
            c["slack_level"] += 1
            c["slack"].fill_(self.SLACK[c["slack_level"]])

    def limits_held(self, c):
        if c is None:
            return
        c["held"] += 1
        if c["held"] >= self.SLACK_RELAX_AFTER and c["slack_level"] > 0:
            c["held"] = 0
            c["slack_level"] -= 1
            c["slack"].fill_(self.SLACK[c["slack_level"]])

    @staticmethod
    def _scratch(geom, img, binning, capacity):
        s = GsScratch()
        s.geom, s.geom_bytes = _ptr(geom), geom.numel()
        s.img, s.img_bytes = _ptr(img), img.numel()
        s.binning, s.binning_bytes = _ptr(binning), binning.numel()
        s.binning_capacity = int(capacity)
        return s

    # ------------------------------------------------------------------ forward
    # The two-phase train step (gs_step_uninstanced): the Adam update of the Gaussians that emitted no instance in this view
    # is launched on a side stream right before the fused backward and runs NEXT TO the backward blend; the per-Gaussian
    # kernel of gs_backward_step (phase 2) waits for it.  From TWO_PHASE_MIN_P Gaussians on (below, the extra launch and
    # the stream hand-offs cost more than the hidden stream saves); GS_TWO_PHASE_STEP=0 switches it off.
    # depth-limited lists for callers that name their cameras (GaussianRasterizer.camera_key); GS_KEYED_LIMITS=0: never
    force_rowwise_entries = False
    KEYED_LIMITS = os.environ.get("GS_KEYED_LIMITS", "1") != "0"
    # region-binned forwards whose verdict is collected later (deferred eager steps, replayed graphs): the status block is
    # written into the pinned host block by the forward's own last kernel (GsScratch.status_host) instead of by a copy
    # command behind it - one launch less on the stream.  False: gs_forward_status as before.
    STATUS_IN_RENDER = True
    # Train step (a fused step is armed): the forward's last kernel (tile_order: the backward's launch order, the camera's next
    # hints and depth bounds, the status block) is issued on the SIDE stream - nothing before the backward blend needs it, so
    # it runs beside the criterion's first kernel instead of in front of it (GsScratch.defer_tile_order +
    # gs_forward_tile_order); the backward waits for it.  False: in line, as the plain forward does.
    SPLIT_TILE_ORDER = True
    REUSE_BUILT = True   # the forward's GsView / GsGaussians structs serve its backward too

    def _side_stream(self, device):
        side = self._side_streams.get(device.index)
        if side is None:
            side = self._side_streams[device.index] = torch.cuda.Stream(device=device)
        return side
    TWO_PHASE = os.environ.get("GS_TWO_PHASE_STEP", "1") != "0"
    TWO_PHASE_MIN_P = 100_000

    # Where the side launch of the two-phase step (gs_step_uninstanced) is issued:
    #   "loss_backward"  by the train step, right before the criterion's backward kernels (Trainer calls
    #                    launch_uninstanced_early): it then runs under ssim_bwd / dwt2_l1_bwd and the backward blend
    #   "ssim_backward"  from inside the criterion's backward, before its last kernel (ssim_bwd)
    #   "raster_backward" at the start of the rasterizer's backward, next to the blend only (round 3)
    # Same box, C3 (gpurun_out/r04/ab4.log): ssim_backward 0.983 ms/step (ssim_bwd 43 -> 52 us beside the stream, blend 0.40),
    # loss_backward 1.024 (the stream starts 20 us earlier, under dwt2_l1_bwd too, and ssim_bwd takes 103 us beside it),
    # raster_backward 1.012 replayed / 1.15 eager (no head start: the blend's 8 160 one-wave workgroups are dealt out first
    # and the stream's workgroups trickle in behind them, phase 2 waits 0.17 ms for it).
    UNINST_AT = "ssim_backward"

    def rasterize_gaussians(self, bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier,
                             cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width,
                             sh, degree, campos, prefiltered, antialiasing, debug, extra=None, fsgs=False, extra_gain=None):
        """The forward (see _rasterize_gaussians); with a fused train step armed it also remembers what an early side launch
        of that step needs (launch_uninstanced_early)."""
        raw = self.raw_activations
        out = self._rasterize_gaussians(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier,
                                        cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width,
                                        sh, degree, campos, prefiltered, antialiasing, debug, extra=extra, fsgs=fsgs,
                                        extra_gain=extra_gain)
        self._early = None
        step = self.fused_step
        if step is not None and not fsgs and means3D.device.type == "cuda" and self.UNINST_AT != "raster_backward":
            P = int(means3D.shape[0])
            if self.TWO_PHASE and P >= self.TWO_PHASE_MIN_P and not step.grad_out[0] and not step.rows_override:
                self._early = dict(args=(bg, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                                         projmatrix, campos, tanfovx, tanfovy, int(image_height), int(image_width),
                                         scale_modifier, degree, antialiasing, debug),
                                   extra=extra, extra_gain=extra_gain,
                                   raw=raw, radii=out[2], geom=out[3], img=out[5], step=step, done=None)
        return out

    def launch_uninstanced_early(self):
        """Issue the side launch of the two-phase step NOW (everything enqueued so far on the current stream - the forward,
        the criterion's forward - is waited for by the side stream; what the caller enqueues next runs beside it).  No-op
        unless the last forward armed it.  The rasterizer's backward then only waits for it."""
        e = self._early
        if e is None or e["done"] is not None:
            return False
        (bg, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tanfovx,
         tanfovy, H, W, scale_modifier, degree, antialiasing, debug) = e["args"]
        device = means3D.device
        self._region_key = None
        built = getattr(self, "_built", None)
        if self.REUSE_BUILT and built is not None and built["geom"] == e["geom"].data_ptr() and built["raw"] == e["raw"]:
            view, g, keep = built["view"], built["g"], built["keep"]
        else:
            keep = []
            view = self._view(keep, device, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, scale_modifier, degree,
                              False, antialiasing, debug)
            g = self._gauss(keep, device, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, e["extra"],
                            raw=e["raw"], extra_gain=e["extra_gain"])
        empty = torch.empty((0,), dtype=torch.uint8, device=device)
        s = self._scratch(e["geom"], e["img"], empty, 0)
        e["done"] = self._launch_uninstanced(device, view, g, e["radii"], s, e["step"])
        e["keep"] = keep
        return True

    def _launch_uninstanced(self, device, view, g, radii, s, step):
        main = torch.cuda.current_stream(device)
        # (a CU-masked side stream - hipExtStreamCreateWithCUMask, 64 / 96 / 128 CUs - was tried in round 4: the two
        #  queues then did not overlap at all, every kernel of the step ran ~10 % slower, step 0.98 -> 1.18-1.20 ms)
        side = self._side_stream(device)
        side.wait_stream(main)   # (the forward has decided overflow / trunc_failed)
        self.api.call("step_uninstanced", C.byref(view), C.byref(g), radii.contiguous().data_ptr(), C.byref(s),
                      C.byref(step), C.c_void_p(side.cuda_stream))
        done = torch.cuda.Event()
        done.record(side)
        self.two_phase_launches += 1
        return done

    def _rasterize_gaussians(self, bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier,
                              cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width,
                              sh, degree, campos, prefiltered, antialiasing, debug, extra=None, fsgs=False, extra_gain=None):
        """= RasterizeGaussiansCUDA (rasterize_points.cu:35-124).

        Returns (num_rendered, color[3,H,W], radii[P] int32, geomBuffer, binningBuffer, imgBuffer,
        invdepth[1,H,W]).  binningBuffer carries its capacity in its length (see _capacity).
        extra [P] (not part of the reference's signature): a 4th per-Gaussian channel blended in the same
        pass (gs_forward_render_x); the tuple then ends with its image [1,H,W].
        fsgs=True: the older generation of FSGS / DNGaussian (gs_forward_render_fsgs): the "invdepth" slot of the
        tuple holds depth = sum depth alpha T and the tuple ends with alpha = sum alpha T, both [1,H,W]."""
        if fsgs and (extra is not None or antialiasing):
            raise RuntimeError("the FSGS rasterizer generation has neither anti-aliasing nor a 4th channel")
        if means3D.ndim != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:58-60
        self._check_device(means3D)
        device = means3D.device
        P, H, W = int(means3D.shape[0]), int(image_height), int(image_width)
        self._region_key = (P, W, H)
        f32 = dict(dtype=torch.float32, device=device)
        u8 = dict(dtype=torch.uint8, device=device)
        # every pixel and every radius is written by the kernels (gs_forward_geometry / gs_forward_render): empty, not
        # zeros - the reference's three fills (rasterize_points.cu:71-75) are only needed for P == 0
        alloc = torch.empty if P != 0 else torch.zeros
        out_color = alloc((NUM_CHANNELS, H, W), **f32)
        out_invdepth = alloc((1, H, W), **f32)
        radii = alloc((P,), dtype=torch.int32, device=device)
        out_extra = None if (extra is None and not fsgs) else alloc((1, H, W), **f32)
        tail = () if out_extra is None else (out_extra,)
        # (the one-shot requests of this forward are taken here, whatever follows: an empty model must not leave them armed)
        raw, self.raw_activations = self.raw_activations, False
        self._raw_backward = raw and P != 0
        sh_rest, self.sh_rest = self.sh_rest, None
        if P == 0:  # rasterize_points.cu:88
            self.camera_key, self.camera_key_limits, self.depth_limit_request = None, False, None
            e = torch.empty((0,), **u8)
            return (0, out_color, radii, e, e.clone(), e.clone(), out_invdepth) + tail
        if sh_rest is not None and not raw:
            raise RuntimeError("split SH rows are served with raw activations only (gsplat_amd.render_raw)")
        if raw and fsgs:
            raise RuntimeError("raw activations do not serve the FSGS rasterizer generation")
        if (extra_gain is not None) != (raw and extra is not None):
            raise RuntimeError("extra_gain goes with a RAW 4th channel (raw activations), and only with it")
        if device.type == "cuda":
            self._join_tile_order(device)   # (a forward whose backward never ran: its hints must be complete before this one reads them)
        cache = self._camera_cache(device, W, H, viewmatrix)
        static = self.static_capacity is not None
        use_order = cache is not None and self.order_hint_on
        request, self.depth_limit_request = self.depth_limit_request, None
        keyed_limits, self.camera_key_limits = self.camera_key_limits and self.KEYED_LIMITS, False
        self.deferred = None
        use_limit = cache is not None and (self.depth_limit_on or request is not None or keyed_limits) and self.tile_cull
        defer = request == "defer"
        # capture-safe mode always passes the limit buffer (+inf = no limit): the pointer is frozen into the graph
        limit = cache["limit"] if use_limit and (static or cache["limit_ok"]) else None

        def scratch_of(binning, capacity, limit):
            s = self._scratch(geom, img, binning, capacity)
            if limit is not None:
                s.tile_depth_limit = limit.data_ptr()
            if static and self.static_step_tag is not None:
                s.step_tag = self.static_step_tag.data_ptr()
            return s

        def render(scratch):
            if use_order and cache["order_ok"]:
                scratch.tile_order_hint = cache["order"].data_ptr()
            # what this view measures becomes the hint of this camera's next visit: written by the forward's last launch
            # straight into the camera's buffers (they are read - as this view's hints - before they are written)
            if use_order:
                scratch.tile_order_out = cache["order"].data_ptr()
            if use_limit:
                scratch.tile_depth_limit_out = cache["limit"].data_ptr()
                scratch.tile_depth_limit_slack = cache["slack"].data_ptr()
            self._render(scratch, fsgs, extra, view, g, out_color, out_invdepth, out_extra, stream)

        def remember(scratch):
            if use_order:
                cache["order_ok"] = True
            if use_limit:
                cache["limit_ok"] = True

        keep = []
        # (depth-limited lists are several times shorter: a size whose FULL lists overfill a region still bins its limited
        #  views by region - C2: 0.146 -> 0.04 ms of binning per step)
        self._region_key = (P, W, H, limit is not None)
        view = self._view(keep, device, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, scale_modifier,
                          degree, prefiltered, antialiasing, debug)
        g = self._gauss(keep, device, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, extra, raw=raw,
                        extra_gain=extra_gain, sh_rest=sh_rest)
        stream = self._stream(device)

        gb, ib, _, _ = self.scratch_bytes(P, W, H, 0)
        geom = torch.empty((gb,), **u8)
        self._raw_geom = geom.data_ptr() if raw else None
        img = torch.empty((ib,), **u8)
        empty = torch.empty((0,), **u8)
        # (the train step's side launch and its backward describe the same view and the same Gaussians: they take these
        #  structs - and the tensors `keep` holds alive - instead of building them again; keyed by this forward's geometry buffer)
        self._built = dict(geom=geom.data_ptr(), view=view, g=g, keep=keep, raw=raw)
        self._last_geom = (geom, P)   # (export_row_mask: the data-parallel step's early mask exchange)

        def new_binning(cap):
            _, _, bb, _ = self.scratch_bytes(P, W, H, cap)
            binning = torch.empty((bb,), **u8)
            self._remember_capacity(binning, cap)
            return binning

        if device.type != "cuda":
            nr = (C.c_int32 * 1)()
            self.api.call("forward_geometry", C.byref(view), C.byref(g), C.byref(scratch_of(empty, 0, None)),
                          radii.data_ptr(), C.cast(nr, C.c_void_p), stream)
            num_rendered = int(nr[0])
            binning = new_binning(num_rendered)
            render(scratch_of(binning, num_rendered, None))
            return (num_rendered, out_color, radii, geom, binning, img, out_invdepth) + tail

        # one pinned status block per device AND stream: two forwards in flight on different streams must not share it
        # (capture-safe mode: one per device, created before the capture - pinned memory cannot be allocated inside).
        # Words: num_rendered, overflow, trunc_failed, 0 (gs_forward_status; gs_forward_geometry writes word 0 alone)
        cur = torch.cuda.current_stream(device)
        pkey = (device.index, "static" if static else cur.cuda_stream)
        status = self._pinned_by_device.get(pkey)
        if status is None:
            status = self._pinned_by_device[pkey] = torch.zeros((16,), dtype=torch.int32).pin_memory()
        self._pinned = status  # (last used: read by bench.py for the instance count of the last view)

        def geometry(limit):
            self.api.call("forward_geometry", C.byref(view), C.byref(g), C.byref(scratch_of(empty, 0, limit)),
                          radii.data_ptr(), status.data_ptr(), stream)

        if view.tile_cull == 2 and self.binning == "auto" and limit is None and not static:
            regions = ((((W + 15) // 16) + 3) // 4) * ((((H + 15) // 16) + 3) // 4)
            if self._capacity_hint > self.REGION_AUTO_MAX * regions:
                view.tile_cull = 1   # long un-limited lists: the LSD path (same lists)
        if view.tile_cull == 2:
            out = self._forward_region(P, W, H, cache, limit, defer, static, status, cur, geom, img, view, g, radii, stream,
                                       new_binning, scratch_of, render, remember)
            if out is not None:
                num_rendered, binning = out
                return (num_rendered, out_color, radii, geom, binning, img, out_invdepth) + tail
            # a region holds more Gaussians than one workgroup sorts: this size goes through the LSD path from now on
            view.tile_cull = 1

        if static:
            # capture-safe: fixed capacity, no host wait, no re-run; the caller reads last_status() after the stream drained
            cap = int(self.static_capacity)
            geometry(limit)
            binning = new_binning(cap)
            s = scratch_of(binning, cap, limit)
            render(s)
            if limit is not None:
                self.depth_limit_stats["used"] += 1  # (at capture time only: replays do not pass here)
            self.api.call("forward_status", C.byref(s), status.data_ptr(), stream)
            remember(s)
            return (cap, out_color, radii, geom, binning, img, out_invdepth) + tail

        for limit in ((limit, None) if limit is not None else (None,)):
            geometry(limit)
            cap = self._capacity_hint if self.optimistic else 0
            if cap > 0 and limit is not None and self._capacity_hint_limited > 0:
                cap = self._capacity_hint_limited
            binning = None
            if cap > 0:
                # Optimistic path: the reference blocks the host on a D2H copy of num_rendered before it can
                # size the binning buffer (rasterizer_impl.cu:284-288) and the GPU idles meanwhile.  Here the
                # binning/blend phase is enqueued at once with a capacity predicted from the previous calls
                # (the kernels read num_rendered on the device); the host then waits only for the geometry
                # phase - the GPU is already sorting and blending - and re-runs the phase in the rare case
                # the prediction was too small.
                ev = torch.cuda.Event()
                ev.record(cur)
                binning = new_binning(cap)
                s = scratch_of(binning, cap, limit)
                render(s)
                ev.synchronize()
            else:
                cur.synchronize()  # the reference's blocking D2H (rasterizer_impl.cu:284)
            num_rendered = int(status[0])
            self._update_hint(num_rendered, limited=limit is not None)
            if binning is None or num_rendered > cap:
                binning = new_binning(num_rendered)
                s = scratch_of(binning, num_rendered, limit)
                render(s)
            if limit is not None:
                # depth-limited lists: the blend has checked that every bounded tile saturated inside the part of its
                # list that is certainly complete; the host has to know before it hands the image out (this wait ends
                # when the blend does, the un-limited path's when the geometry phase does)
                self.depth_limit_stats["used"] += 1
                if defer:
                    # the caller collects the verdict later (take_deferred): its own status block, from a small ring -
                    # a block is re-used only after eight further deferred forwards
                    ring = self._status_ring.setdefault(device.index, [[], 0])
                    if len(ring[0]) < 8:
                        ring[0].append(torch.zeros((16,), dtype=torch.int32).pin_memory())
                    block = ring[0][ring[1] % len(ring[0])]
                    ring[1] += 1
                    self.api.call("forward_status", C.byref(s), block.data_ptr(), stream)
                    done = torch.cuda.Event()
                    done.record(cur)
                    self.deferred = dict(status=block, event=done, cache=cache, limited=True)
                    remember(s)
                    return (num_rendered, out_color, radii, geom, binning, img, out_invdepth) + tail
                self.api.call("forward_status", C.byref(s), status.data_ptr(), stream)
                cur.synchronize()
                if int(status[2]) != 0:  # some tile needed entries that were cut: forget the limits, do the view again
                    self.depth_limit_stats["failed"] += 1
                    cache["limit_ok"] = False
                    self.limits_failed(cache)
                    del binning
                    continue
                self.limits_held(cache)
            remember(s)
            return (num_rendered, out_color, radii, geom, binning, img, out_invdepth) + tail

    REGION_MAX_ENTRIES = 16384  # csrc/gs_common.h RG_MAX_ENTRIES
    REGION_AUTO_MAX = 13000     # binning = "auto": instances of capacity per region above which un-limited forwards take the LSD path

    def _forward_region(self, P, W, H, cache, limit, defer, static, status, cur, geom, img, view, g, radii, stream,
                        new_binning, scratch_of, render, remember):
        """The forward with region binning (GsView.tile_cull = 2, csrc/gs_regionbin.hip).  The region buckets live in the
        binning buffer, so it is allocated BEFORE the geometry phase, from the capacity the previous views needed; the
        instance count is known once the lists are built (gs_forward_bin copies the status words out), which is what
        the host waits for - the blend is already running then.  Too small a capacity (or a bucket that overflowed)
        sets the overflow flag: nothing valid was produced and the view is rendered again with what the status asks for.
        -> (num_rendered, binning buffer), or None when a region holds more Gaussians than one workgroup can sort."""
        regions = ((((W + 15) // 16) + 3) // 4) * ((((H + 15) // 16) + 3) // 4)

        def needed(st):
            return max(int(st[0]), int(st[3]) * regions)

        def geometry(s):
            self.api.call("forward_geometry", C.byref(view), C.byref(g), C.byref(s), radii.data_ptr(), None, stream)

        if static:
            cap = max(int(self.static_capacity), regions)
            binning = new_binning(cap)
            s = scratch_of(binning, cap, limit)
            geometry(s)
            self.api.call("forward_bin", C.byref(view), C.byref(g), C.byref(s), None, stream)
            if self.STATUS_IN_RENDER:
                s.status_host = status.data_ptr()   # the forward's last kernel delivers the status block (+ tag) itself
            render(s)   # (tile_order stays in line here: inside a captured graph the extra cross-stream edge costs more than the
            #              kernel it takes off the chain - C3 replay 0.96 -> 1.00 ms; the eager step below gains 6-10 us)
            if limit is not None:
                self.depth_limit_stats["used"] += 1
            if not self.STATUS_IN_RENDER:
                self.api.call("forward_status", C.byref(s), status.data_ptr(), stream)
            remember(s)
            return cap, binning

        limits = (limit, None) if limit is not None else (None,)
        for limit in limits:
            limited = limit is not None
            hint = self._capacity_hint_limited if (limited and self._capacity_hint_limited > 0) else self._capacity_hint
            cap = hint if hint > 0 else max(8 * P, 1 << 20)
            for attempt in range(6):
                cap = max(cap, regions)
                binning = new_binning(cap)
                s = scratch_of(binning, cap, limit)
                geometry(s)
                if limited and defer:
                    # the caller collects the verdict (capacity AND limits) later: no host wait at all in this forward
                    self.api.call("forward_bin", C.byref(view), C.byref(g), C.byref(s), None, stream)
                    ring = self._status_ring.setdefault(cur.device.index if hasattr(cur, "device") else 0, [[], 0])
                    if len(ring[0]) < 8:
                        ring[0].append(torch.zeros((16,), dtype=torch.int32).pin_memory())
                    block = ring[0][ring[1] % len(ring[0])]
                    ring[1] += 1
                    if self.STATUS_IN_RENDER:   # the forward's last kernel writes the status words into the pinned block
                        s.status_host = block.data_ptr()
                    split = self._split_tile_order(s)
                    render(s)
                    self.depth_limit_stats["used"] += 1
                    if not self.STATUS_IN_RENDER:
                        self.api.call("forward_status", C.byref(s), block.data_ptr(), stream)
                    if split:
                        done = self._tile_order_on_side(view, s, cur)   # (the status block arrives with that kernel)
                    else:
                        done = torch.cuda.Event()
                        done.record(cur)
                    self.deferred = dict(status=block, event=done, cache=cache, regions=regions, limited=True, size=(P, W, H, True))
                    remember(s)
                    return cap, binning
                self.api.call("forward_bin", C.byref(view), C.byref(g), C.byref(s), status.data_ptr(), stream)
                ev = torch.cuda.Event()
                ev.record(cur)
                render(s)
                ev.synchronize()  # the lists are built (the blend is running): did they fit?
                st = tuple(int(x) for x in status[:4])
                if st[1] == 0:
                    break
                if st[3] > self.REGION_MAX_ENTRIES:
                    self._region_off.add((P, W, H, limited))
                    return None
                cap = int(needed(st) * 1.25) + 4096
                del binning
            else:
                raise RuntimeError("region binning: the capacity did not settle (status %r)" % (st,))
            self._update_hint(needed(st), limited=limited)
            num_rendered = st[0]
            if limited:
                # depth-limited lists: the blend checks that every bounded tile saturated inside the part of its list
                # that is certainly complete; the host has to know before it hands the image out
                self.depth_limit_stats["used"] += 1
                self.api.call("forward_status", C.byref(s), status.data_ptr(), stream)
                cur.synchronize()
                if int(status[2]) != 0:
                    self.depth_limit_stats["failed"] += 1
                    cache["limit_ok"] = False
                    self.limits_failed(cache)
                    del binning
                    continue
                self.limits_held(cache)
            remember(s)
            return num_rendered, binning

    def _split_tile_order(self, scratch):
        """Arm GsScratch.defer_tile_order for the render about to be issued?  Only inside a train step (a fused step is armed:
        its backward is what waits for the side launch) and only with the status delivered by that kernel."""
        if not (self.SPLIT_TILE_ORDER and self.fused_step is not None and self.STATUS_IN_RENDER):
            return False
        scratch.defer_tile_order = 1
        return True

    def _tile_order_on_side(self, view, scratch, main):
        """gs_forward_tile_order on the side stream, ordered behind the render just issued on `main`; the event it returns (also
        kept in _tile_order_done) is what the backward - and any later forward - waits for."""
        side = self._side_stream(main.device)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        # (no `with torch.cuda.stream(side)`: the call takes the stream, the event records on it - the context switch alone
        #  is ~20 us of host time per step)
        self.api.call("forward_tile_order", C.byref(view), C.byref(scratch), C.c_void_p(side.cuda_stream))
        done = torch.cuda.Event()
        done.record(side)
        self._tile_order_done = done
        return done

    def _join_tile_order(self, device):
        """Order the current stream behind a tile_order still pending on the side stream (no-op otherwise)."""
        done, self._tile_order_done = getattr(self, "_tile_order_done", None), None
        if done is not None:
            torch.cuda.current_stream(device).wait_event(done)

    def _render(self, scratch, fsgs, extra, view, g, out_color, out_invdepth, out_extra, stream):
        if fsgs:
            self.api.call("forward_render_fsgs", C.byref(view), C.byref(g), C.byref(scratch), out_color.data_ptr(),
                          out_invdepth.data_ptr(), out_extra.data_ptr(), stream)
        elif extra is None:
            self.api.call("forward_render", C.byref(view), C.byref(g), C.byref(scratch), out_color.data_ptr(),
                          out_invdepth.data_ptr(), stream)
        else:
            self.api.call("forward_render_x", C.byref(view), C.byref(g), C.byref(scratch), out_color.data_ptr(),
                          out_invdepth.data_ptr(), out_extra.data_ptr(), stream)

    # ------------------------------------------------------------------ backward
    def rasterize_gaussians_backward(self, bg, means3D, radii, colors_precomp, opacities, scales, rotations,
                                     scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy,
                                     dL_dout_color, dL_dout_invdepth, sh, degree, campos, geomBuffer, R,
                                     binningBuffer, imgBuffer, antialiasing, debug, extra=None, dL_dout_extra=None,
                                     fsgs=False, extra_gain=None):
        """= RasterizeGaussiansBackwardCUDA (rasterize_points.cu:126-223).

        Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
        dL_drotations) (+ dL_dextra [P] when the forward blended a 4th channel).
        fsgs=True: dL_dout_invdepth is the depth image gradient and dL_dout_extra the alpha image gradient."""
        self._check_device(means3D)
        device = means3D.device
        P = int(means3D.shape[0])
        H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])
        self._region_key = None
        if device.type == "cuda":
            self._join_tile_order(device)   # (the blend backward reads the launch order that kernel writes)
        M = int(sh.shape[1]) if (sh is not None and sh.numel() != 0) else 0
        f32 = dict(dtype=torch.float32, device=device)
        # every row is written by gs_backward (culled rows become 0): empty, not zeros
        alloc = torch.empty if P != 0 else torch.zeros
        arena, self.grad_arena = self.grad_arena, None
        step, self.fused_step = self.fused_step, None
        raw, self._raw_backward = self._raw_backward, False
        if raw and step is None and self._raw_geom is not None and self._raw_geom != geomBuffer.data_ptr():
            # this backward belongs to an EARLIER forward on activated values; the raw forward in between (an evaluation render
            # under no_grad, say) has no backward of its own
            raw = False
        sh_rest, self.sh_rest = self.sh_rest, None
        if sh_rest is not None and (step is None or not raw or not step.grad_out_rest):
            raise RuntimeError("split SH rows: the backward is gs_backward_step's gradients-out form with grad_out_rest")
        if raw and (step is None or P == 0):
            raise RuntimeError("a forward on raw activations must be followed by the fused train-step backward")
        if step is not None and P != 0:
            if fsgs:
                raise RuntimeError("the fused train-step backward does not serve the FSGS rasterizer generation")
            built = getattr(self, "_built", None)
            if self.REUSE_BUILT and built is not None and built["geom"] == geomBuffer.data_ptr() and built["raw"] == raw:
                view, g, keep = built["view"], built["g"], built["keep"]   # (this view's forward built them: same pointers)
            else:
                keep = []
                view = self._view(keep, device, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, scale_modifier,
                                  degree, False, antialiasing, debug)
                g = self._gauss(keep, device, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, extra,
                                raw=raw, extra_gain=extra_gain, sh_rest=sh_rest)
            if extra is not None:
                dL_dout_extra = torch.zeros((1, H, W), **f32) if dL_dout_extra is None else _prep(dL_dout_extra, device)
            dL_dout_color = _prep(dL_dout_color, device)
            dL_dout_invdepth = _prep(dL_dout_invdepth, device)
            _, _, _, wsb = self.scratch_bytes(P, W, H, R)
            # (the backward clears only the rows of Gaussians that emitted instances; a probe that reads the rows gets zeros
            # for the others too)
            two_phase = self.TWO_PHASE and P >= self.TWO_PHASE_MIN_P and not step.grad_out[0]
            if self.keep_workspace or step.rows_override:
                ws = (torch.zeros if self.keep_workspace else torch.empty)((wsb,), dtype=torch.uint8, device=device)
                if self.keep_workspace:
                    self.last_workspace = ws
                step.rows_clean = 0
            else:
                # one persistent workspace per size: the chain kernel zeroes every row it consumes (GsStepState.rows_clean),
                # so the rows are clean again after every step and no clear launch runs.  rows_clean = 1 - one clear launch,
                # then clean again - whenever the host is not SURE of that state (first use, a call that raised)
                key = (device.index, wsb)
                ent = self._rows_ws.get(key)
                if ent is None:
                    if len(self._rows_ws) >= 4:   # (sizes of a model that has since been densified: nothing replays them,
                        self._rows_ws.clear()    #  a captured graph is keyed by the number of Gaussians)
                    ent = self._rows_ws[key] = [torch.zeros((wsb,), dtype=torch.uint8, device=device), True]
                ws = ent[0]
                # (a launch captured into a hipGraph is replayed with this flag frozen: `rows_epoch` counts the calls that may
                #  have left rows dirty, GraphedStep keys its graphs by it and captures again - after an eager step that cleaned)
                step.rows_clean = 2 if ent[1] else 1
                if not ent[1]:
                    self.rows_epoch += 1
                ent[1] = False      # (until the call below has returned: a failed launch leaves them in an unknown state)
            s = self._scratch(geomBuffer, imgBuffer, binningBuffer, self._capacity_for(binningBuffer, P, W, H, R))
            early, self._early = self._early, None
            if two_phase:
                if early is not None and early["done"] is not None and early["geom"].data_ptr() == geomBuffer.data_ptr() \
                        and early["step"] is step:
                    done = early["done"]          # (issued by launch_uninstanced_early, under the criterion's backward)
                else:
                    done = self._launch_uninstanced(device, view, g, radii, s, step)
                self._uninst_done = done          # (kept alive until the next step replaces it)
                step.phase = 2                    # gs_backward_step: the Gaussians with instances only ...
                step.phase1_done = done.cuda_event  # ... its per-Gaussian kernel behind the side launch
            try:
                if extra is not None:
                    self.api.call("backward_step_x", C.byref(view), C.byref(g), radii.contiguous().data_ptr(), C.byref(s), int(R),
                                  dL_dout_color.data_ptr(), _ptr(dL_dout_invdepth), dL_dout_extra.data_ptr(), C.byref(step),
                                  _ptr(ws), ws.numel(), self._stream(device))
                else:
                    self.api.call("backward_step", C.byref(view), C.byref(g), radii.contiguous().data_ptr(), C.byref(s), int(R),
                                  dL_dout_color.data_ptr(), _ptr(dL_dout_invdepth), C.byref(step), _ptr(ws), ws.numel(),
                                  self._stream(device))
            except Exception:
                self.rows_epoch += 1   # (the persistent rows may hold sums now: graphs that skip the clear are stale)
                raise
            if step.rows_clean:
                self._rows_ws[(device.index, wsb)][1] = True
            return (None,) * (8 if extra is None else 9)

        def out(name, shape):
            t = None if arena is None else arena.get(name)
            if t is not None and tuple(t.shape) == tuple(shape) and t.is_contiguous() and t.device == device \
                    and t.dtype == torch.float32:
                return t
            return alloc(shape, **f32)
        dL_dmeans3D = out("means3D", (P, 3))
        dL_dmeans2D = alloc((P, 3), **f32)
        # gradients of inputs that were not given ("absent" colours / covariances) are not produced: the reference
        # returns zero tensors for them (rasterize_points.cu:163-178) which autograd then drops
        has_colors = colors_precomp is not None and colors_precomp.numel() != 0
        has_cov = cov3D_precomp is not None and cov3D_precomp.numel() != 0
        dL_dcolors = alloc((P, NUM_CHANNELS), **f32) if has_colors else None
        dL_dopacity = alloc((P, 1), **f32)
        dL_dcov3D = alloc((P, 6), **f32) if has_cov else None
        dL_dsh = out("sh", (P, M, 3))
        dL_dscales = alloc((P, 3), **f32)
        dL_drotations = alloc((P, 4), **f32)
        ret = (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
        dL_dextra = None
        if extra is not None:
            dL_dextra = alloc((P,), **f32)
            ret = ret + (dL_dextra,)
        if P == 0:
            return ret
        keep = []
        view = self._view(keep, device, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, scale_modifier,
                          degree, False, antialiasing, debug)
        g = self._gauss(keep, device, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, extra)
        stream = self._stream(device)
        dL_dout_color = _prep(dL_dout_color, device)
        dL_dout_invdepth = _prep(dL_dout_invdepth, device)
        radii = radii.contiguous()
        _, _, _, wsb = self.scratch_bytes(P, W, H, R)
        ws = (torch.zeros if self.keep_workspace else torch.empty)((wsb,), dtype=torch.uint8, device=device)
        if self.keep_workspace:
            self.last_workspace = ws
        cap = self._capacity_for(binningBuffer, P, W, H, R)
        s = self._scratch(geomBuffer, imgBuffer, binningBuffer, cap)
        grads = GsGrads()
        grads.dL_dmeans3D, grads.dL_dmeans2D = dL_dmeans3D.data_ptr(), dL_dmeans2D.data_ptr()
        grads.dL_dsh = _ptr(dL_dsh)
        grads.dL_dcolors, grads.dL_dopacity = _ptr(dL_dcolors), dL_dopacity.data_ptr()
        grads.dL_dscales, grads.dL_drotations = dL_dscales.data_ptr(), dL_drotations.data_ptr()
        grads.dL_dcov3D = _ptr(dL_dcov3D)
        if g.scales is None:
            dL_dscales.zero_()
            dL_drotations.zero_()
        if fsgs:
            zeros = None
            if dL_dout_invdepth is None or _prep(dL_dout_extra, device) is None:
                zeros = torch.zeros((1, H, W), **f32)
            gd = dL_dout_invdepth if dL_dout_invdepth is not None else zeros
            ga = _prep(dL_dout_extra, device)
            ga = ga if ga is not None else zeros
            self.api.call("backward_fsgs", C.byref(view), C.byref(g), radii.data_ptr(), C.byref(s), int(R),
                          dL_dout_color.data_ptr(), gd.data_ptr(), ga.data_ptr(), C.byref(grads), _ptr(ws), ws.numel(),
                          stream)
        elif extra is None:
            self.api.call("backward", C.byref(view), C.byref(g), radii.data_ptr(), C.byref(s), int(R),
                          dL_dout_color.data_ptr(), _ptr(dL_dout_invdepth), C.byref(grads), _ptr(ws), ws.numel(), stream)
        else:
            grads.dL_dextra = dL_dextra.data_ptr()
            dL_dout_extra = _prep(dL_dout_extra, device)
            if dL_dout_extra is None:
                dL_dout_extra = torch.zeros((1, H, W), **f32)
            self.api.call("backward_x", C.byref(view), C.byref(g), radii.data_ptr(), C.byref(s), int(R),
                          dL_dout_color.data_ptr(), _ptr(dL_dout_invdepth), dL_dout_extra.data_ptr(), C.byref(grads),
                          _ptr(ws), ws.numel(), stream)
        return ret

    def backward_from_rows(self, rows, bg, means3D, radii, colors_precomp, opacities, scales, rotations, scale_modifier,
                           cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, H, W, sh, degree, campos, geomBuffer,
                           antialiasing, depth_mode=0):
        """Stage 2 of the backward alone (gs_backward_from_rows): the per-Gaussian chain rule applied to GIVEN sums of the
        blend backward, rows [P,16] float64.  Same return tuple as rasterize_gaussians_backward.  Parity tests only."""
        self._check_device(means3D)
        device = means3D.device
        P = int(means3D.shape[0])
        M = int(sh.shape[1]) if (sh is not None and sh.numel() != 0) else 0
        f32 = dict(dtype=torch.float32, device=device)
        has_colors = colors_precomp is not None and colors_precomp.numel() != 0
        has_cov = cov3D_precomp is not None and cov3D_precomp.numel() != 0
        out = dict(means2D=torch.empty((P, 3), **f32), colors=torch.empty((P, 3), **f32) if has_colors else None,
                   opacity=torch.empty((P, 1), **f32), means3D=torch.empty((P, 3), **f32),
                   cov3D=torch.empty((P, 6), **f32) if has_cov else None,
                   sh=torch.empty((P, M, 3), **f32) if M else None, scales=torch.zeros((P, 3), **f32),
                   rotations=torch.zeros((P, 4), **f32))
        keep = []
        view = self._view(keep, device, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, scale_modifier, degree,
                          False, antialiasing, False)
        g = self._gauss(keep, device, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp)
        grads = GsGrads()
        grads.dL_dmeans3D, grads.dL_dmeans2D = out["means3D"].data_ptr(), out["means2D"].data_ptr()
        grads.dL_dsh, grads.dL_dcolors, grads.dL_dopacity = _ptr(out["sh"]), _ptr(out["colors"]), out["opacity"].data_ptr()
        grads.dL_dscales, grads.dL_drotations = out["scales"].data_ptr(), out["rotations"].data_ptr()
        grads.dL_dcov3D = _ptr(out["cov3D"])
        rows = rows.to(device=device, dtype=torch.float64).contiguous()
        assert rows.shape == (P, 16)
        s = self._scratch(geomBuffer, torch.empty(0, dtype=torch.uint8, device=device),
                          torch.empty(0, dtype=torch.uint8, device=device), 0)
        _, _, _, wsb = self.scratch_bytes(P, W, H, 0)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=device)
        self.api.call("backward_from_rows", C.byref(view), C.byref(g), radii.contiguous().data_ptr(), C.byref(s),
                      rows.data_ptr(), int(depth_mode), C.byref(grads), ws.data_ptr(), ws.numel(), self._stream(device))
        return (out["means2D"], out["colors"], out["opacity"], out["means3D"], out["cov3D"], out["sh"], out["scales"],
                out["rotations"])

    # ------------------------------------------------------------------ markVisible
    def export_row_mask(self, out):
        """out [P] uint8 <- 1 where the Gaussian emitted instances in the LAST forward (gs_export_row_mask: what the
        data-parallel backward later writes into GsStepState.grad_mask).  False when there is no such forward to ask."""
        last = getattr(self, "_last_geom", None)
        if last is None or last[0].device != out.device or last[1] != out.numel() or not hasattr(self.api, "_export_row_mask"):
            return False
        geom, P = last
        empty = torch.empty((0,), dtype=torch.uint8, device=geom.device)
        s = self._scratch(geom, empty, empty, 0)
        self.api.call("export_row_mask", C.byref(s), P, out.data_ptr(), self._stream(geom.device))
        return True

    def mark_visible(self, means3D, viewmatrix, projmatrix):
        """= markVisible (rasterize_points.cu:225-244)."""
        self._check_device(means3D)
        device = means3D.device
        P = int(means3D.shape[0])
        present = torch.zeros((P,), dtype=torch.bool, device=device)
        if P != 0:
            m, vm, pm = (_prep(x, device) for x in (means3D, viewmatrix, projmatrix))
            self.api.call("mark_visible", P, m.data_ptr(), vm.data_ptr(), pm.data_ptr(), present.data_ptr(),
                          self._stream(device))
        return present

    # ------------------------------------------------------------------ parity exports
    def export_state(self, P, W, H, R, geom, binning, img):
        """Plain-array copies of the forward's internal state (tests / debugging only)."""
        device = geom.device
        cap = self._capacity_for(binning, P, W, H, R)
        s = self._scratch(geom, img, binning, cap)
        f32 = dict(dtype=torch.float32, device=device)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        out = dict(
            depths=torch.zeros((P,), **f32), means2D=torch.zeros((P, 2), **f32), cov3D=torch.zeros((P, 6), **f32),
            conic_opacity=torch.zeros((P, 4), **f32), rgb=torch.zeros((P, 3), **f32),
            clamped=torch.zeros((P, 3), dtype=torch.uint8, device=device),
            tiles_touched=torch.zeros((P,), dtype=torch.int32, device=device),
            point_offsets=torch.zeros((P,), dtype=torch.int32, device=device),
            keys_sorted=torch.zeros((R,), dtype=torch.int64, device=device),
            point_list=torch.zeros((R,), dtype=torch.int32, device=device),
            final_T=torch.zeros((H, W), **f32), n_contrib=torch.zeros((H, W), dtype=torch.int32, device=device),
            ranges=torch.zeros((T, 2), dtype=torch.int32, device=device))
        st = self._stream(device)
        o = out
        self.api.call("export_geom", C.byref(s), P, _ptr(o["depths"]), _ptr(o["means2D"]), _ptr(o["cov3D"]),
                      _ptr(o["conic_opacity"]), _ptr(o["rgb"]), _ptr(o["clamped"]), _ptr(o["tiles_touched"]),
                      _ptr(o["point_offsets"]), st)
        from .capi import GsError
        try:
            self.api.call("export_binning", C.byref(s), R, _ptr(o["keys_sorted"]), _ptr(o["point_list"]), st)
        except GsError as e:
            if e.code != -5:
                raise
            # lists built by region binning: keys rebuilt from ranges[] (tile lists lie in point_list in no particular order)
            self.api.call("export_binning_region", C.byref(s), W, H, R, _ptr(o["keys_sorted"]), _ptr(o["point_list"]), st)
        self.api.call("export_img", C.byref(s), W, H, _ptr(o["final_T"]), _ptr(o["n_contrib"]), _ptr(o["ranges"]), st)
        if device.type == "cuda":
            torch.cuda.current_stream(device).synchronize()
        return out
