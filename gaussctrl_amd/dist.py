"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm) / xGMI.

The hot path shards BY VIEW: renders are independent per camera and a non-reference view's denoise
trajectory depends only on its own latent plus the 4 reference views' K/V.  Partitioning: view v belongs to rank
v % world_size; Gaussian parameters and network weights are replicated.

Collectives:
  * gradient reduction of the N x 59 fp32 Gaussian parameters after each training render: ONE flat all-reduce
    (236 / 472 / 944 MB at 1 / 2 / 4 M Gaussians) instead of a per-tensor DDP bucket walk -- xGMI rings are
    per-link bound, so few large messages;
  * all-gather of the edited images at the end of edit_images (3 MB per view) so every rank trains on all views;
  * the reference K/V bank, three ways: (a) `RefShard` + DenoisePipeline.begin_ref_bank_sharded -- the reference trajectory itself is
    sharded by sample and every cross-view attention layer ALL-GATHERS K / V^T (north_star's collective in its literal form; nobody
    carries extra work; `bench.py --ref-mode allgather`, GaussCtrlPipelineConfig.ref_bank_allgather); (b) `RefBankStream` -- an owner
    rank (rotating per scene in bench.py, the default there: `--ref-mode rotate`) computes the trajectory and posts each DDIM step's
    K / V^T as ONE flat async broadcast (20 messages of ~0.5 GB per scene: ring collectives over xGMI are per-link bound) that rides
    under the other ranks' denoise kernels, views load-balanced around the owner (`shard_views_balanced`); (c) replicated (every rank
    runs the 4-view trajectory itself: no data-path collective -- the plugin's default, ref_bank_owner = -1, and the A/B
    `--ref-mode replicate`).
These helpers are device-agnostic so the N>1 logic is covered by world_size-2 gloo tests on CPU."""
from __future__ import annotations

import torch


def shard_views(n_views: int, world_size: int, rank: int) -> list[int]:
    return [v for v in range(n_views) if v % world_size == rank]


REF_VIEW_EQUIV = 4.0      # the 4-view reference trajectory (CFG batch 8, 20 steps) costs about what editing 4 views costs


def shard_views_balanced(n_views: int, world_size: int, rank: int, owner: int = -1, ref_equiv: float = REF_VIEW_EQUIV) -> list[int]:
    """View sharding that accounts for the reference trajectory: the rank that computes a scene's (or the next scene's) reference bank
    while the others edit (`owner`) gets ref_equiv fewer views, everybody else correspondingly more -- so that all ranks finish a
    scene together instead of N - 1 ranks waiting for the owner.  owner < 0 (replicated bank / single rank): plain v % N sharding.
    Deterministic, contiguous in rank order, the same on every rank."""
    if world_size <= 1 or owner < 0:
        return shard_views(n_views, world_size, rank)
    share = (n_views + ref_equiv) / world_size                      # work per rank in view units
    n_owner = max(0, min(n_views, int(round(share - ref_equiv))))
    rest, others = n_views - n_owner, world_size - 1
    counts = []
    for r in range(world_size):
        if r == owner:
            counts.append(n_owner)
        else:
            j = r if r < owner else r - 1                           # index among the non-owners
            counts.append(rest // others + (1 if j < rest % others else 0))
    start = sum(counts[:rank])
    return list(range(start, start + counts[rank]))


def split_chunks(views: list, n_chunks: int, chunk_size: int) -> list[list]:
    """a rank's views as exactly n_chunks chunks of <= chunk_size views, sizes as even as possible (ranks run their chunks in lock
    step: the gradient all-reduce after every chunk is a rendez-vous), empty chunks last"""
    n = len(views)
    assert n <= n_chunks * chunk_size, (n, n_chunks, chunk_size)
    base, extra = divmod(n, n_chunks)
    out, o = [], 0
    for j in range(n_chunks):
        k = base + (1 if j < extra else 0)
        out.append(list(views[o:o + k])); o += k
    return out


def allreduce_gradients(params, world_size: int, average: bool = True) -> None:
    """Flat all-reduce of every .grad in `params` (in place; a gather copy + a blocking collective + a copy back).  NOT the path of a normal
    training step (that is FlatGrads: the backward writes the buffer RCCL reduces in place); it serves the cases where autograd owns the
    .grad tensors: gradient accumulation over several steps, a crop box during training, parameters outside the six leaf tensors."""
    if world_size <= 1:
        return
    import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    if average:
        flat /= world_size
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()


def allgather_view_images(local: dict, n_views: int, world_size: int, rank: int, shape, device) -> dict:
    """local: {view index -> image tensor of `shape`} for this rank's views -> {view -> image} for ALL views."""
    if world_size <= 1:
        return dict(local)
    import torch.distributed as dist
    per = (n_views + world_size - 1) // world_size
    mine = torch.zeros((per,) + tuple(shape), device=device)
    for j, v in enumerate(shard_views(n_views, world_size, rank)):
        mine[j] = local[v]
    bufs = [torch.empty_like(mine) for _ in range(world_size)]
    dist.all_gather(bufs, mine)
    out = {}
    for r in range(world_size):
        for j, v in enumerate(shard_views(n_views, world_size, r)):
            out[v] = bufs[r][j]
    return out


def _bank_steps(store: dict) -> list:
    return sorted({k[0] for k in store})


def _step_meta(bank, st: int) -> list:
    """[(layer key, K shape, K row stride, V^T shape, dtype)] of one DDIM step, in a rank-independent order."""
    items = sorted(((key, kv) for key, kv in bank.store.items() if key[0] == st), key=lambda kv: str(kv[0]))
    return [(key[1], tuple(k.shape), k.stride(1), tuple(vt.shape), str(k.dtype)) for key, (k, vt) in items]


def _pack_step(bank, st: int, layers: list) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for layer, *_ in layers for t in (bank.store[(st, layer)][0].contiguous(), bank.store[(st, layer)][1])])


def _step_numel(layers: list) -> int:
    return sum(int(torch.Size(ks).numel()) + int(torch.Size(vs).numel()) for _, ks, _, vs, _ in layers)


def _unpack_step(bank, st: int, layers: list, flat: torch.Tensor) -> None:
    o = 0
    for layer, ks, kstride, vs, dt in layers:
        nk, nv = int(torch.Size(ks).numel()), int(torch.Size(vs).numel())
        buf = torch.empty(ks[0], ks[1], kstride, dtype=flat.dtype, device=flat.device)      # row stride of the Q|K buffer
        k = buf[..., kstride - ks[2]:]
        k.copy_(flat[o:o + nk].view(ks)); o += nk
        vt = flat[o:o + nv].view(vs).clone(); o += nv
        bank.store[(st, layer)] = (k, vt)


def broadcast_ref_bank(bank, src: int, world_size: int, rank: int, device=None, steps=None):
    """Broadcast a finished RefBank (gaussctrl_amd.sd.unet.RefBank) from rank `src` to every rank, one flat buffer per DDIM step.

    On `src`, `bank` holds {(step, layer): (K [R,L,C] (a column slice of the Q|K buffer), V^T [R,C,Lp])}.  The other ranks pass an
    empty RefBank and get the same keys; K is re-materialised with the row stride the attention kernel expects (2 C: the bank K
    is read with the live K's leading dimension).  `steps` restricts the call to some DDIM steps."""
    if world_size <= 1:
        return bank
    import torch.distributed as dist
    meta = [None]
    if rank == src:
        todo = _bank_steps(bank.store) if steps is None else list(steps)
        meta[0] = [(st, _step_meta(bank, st)) for st in todo]
    dist.broadcast_object_list(meta, src=src)
    for st, layers in meta[0]:
        dtype = getattr(torch, layers[0][4].split(".")[-1])
        flat = _pack_step(bank, st, layers) if rank == src else torch.empty(_step_numel(layers), dtype=dtype, device=device)
        dist.broadcast(flat, src=src)
        if rank != src:
            _unpack_step(bank, st, layers, flat)
    if rank != src:
        bank.mode = "use"
    return bank


class RefShard:
    """Sharding of the reference trajectory itself (SURVEY.md 8e / north_star: "RCCL all-gather of reference-view K/V over xGMI").
    The trajectory's network batch is 2 CFG halves x 4 reference frames = 8 samples, sample s = half * 4 + frame; rank r runs the samples
    s % world == r (world in {2, 4, 8}: world 2 -> both halves of frames {r, r + 2}; 4 -> both halves of frame r; 8 -> one sample).
    Every cross-view attention layer needs K / V^T of all four references of a half and the edit chunks later need all eight: one
    all-gather per layer (K and V^T packed into one message) gives both.  When a rank holds a single half (world 8) the CFG
    combination needs the partner half's eps: one small all-gather per DDIM step.  Compared with an owner rank + broadcast
    (RefBankStream) nobody carries ~4 views' worth of extra work, so views shard plainly as v % world."""

    def __init__(self, world_size: int, rank: int, group=None):
        assert world_size in (2, 4, 8), "the 8 reference samples shard over 2, 4 or 8 ranks"
        self.world, self.rank, self.group = world_size, rank, group
        self.samples = [s for s in range(8) if s % world_size == rank]
        self.frames = sorted({s % 4 for s in self.samples})
        self.halves = sorted({s // 4 for s in self.samples})
        self.half_base = 4 * self.halves[0]
        self.per_rank = len(self.samples)
        self._host_sync = None

    def _sync_if_needed(self, t):
        if self._host_sync is None:
            import torch.distributed as dist
            self._host_sync = dist.get_backend(self.group) != "nccl"     # c10d's NCCL work orders itself after the current stream
        if self._host_sync and t.is_cuda:
            torch.cuda.current_stream().synchronize()                    # gloo reads / writes the buffers from the host side

    def _all_gather(self, flat):
        import torch.distributed as dist
        out = torch.empty(self.world * flat.numel(), dtype=flat.dtype, device=flat.device)
        self._sync_if_needed(flat)
        dist.all_gather_into_tensor(out, flat, group=self.group)
        self._sync_if_needed(flat)
        return out.view(self.world, -1)

    def gather_kv(self, k, vt):
        """k [S_loc, L, C] (a column slice of the Q | K buffer), vt [S_loc, C, Lp] of this rank's samples -> (K [8, L, C] with the row
        stride the attention kernel reads the live K with, V^T [8, C, Lp]) in sample order."""
        S, L, C = k.shape
        Lp = vt.shape[-1]
        ld = k.stride(1)
        nk = S * L * C
        g = self._all_gather(torch.cat([k.reshape(-1) if k.is_contiguous() else k.contiguous().reshape(-1), vt.reshape(-1)]))
        # rank r's slot j is sample j * world + r: [world, S] -> [S, world] = sample order
        kf = g[:, :nk].view(self.world, S, L, C).transpose(0, 1).reshape(8, L, C)
        vf = g[:, nk:].view(self.world, S, C, Lp).transpose(0, 1).reshape(8, C, Lp).contiguous()
        buf = torch.empty(8, L, ld, dtype=k.dtype, device=k.device)
        kr = buf[..., ld - C:]
        kr.copy_(kf)
        return kr, vf

    def gather_eps_pairs(self, eps):
        """eps [S_loc, h, w, c] of this rank's single-half samples -> [2 * S_loc, h, w, c] = [unconditional ; conditional] of the same frames"""
        g = self._all_gather(eps.reshape(-1)).view((self.world, self.per_rank) + tuple(eps.shape[1:])).transpose(0, 1).reshape((8,) + tuple(eps.shape[1:]))
        idx = [f for f in self.frames] + [4 + f for f in self.frames]
        return g[idx].contiguous()


MAX_INFLIGHT = 3        # async reference-bank broadcasts in flight (owner: packed copies kept; others: receive buffers posted)


class RefBankStream:
    """SURVEY.md 8e collective 1 as a stream.  The owner rank advances the 4-view reference trajectory and posts each finished DDIM
    step's K / V^T (one flat message, ~0.5 GB at SD1.5 / 512x512) as an ASYNC broadcast; every other rank posts the matching
    receive.  `advance(n)` moves the stream n DDIM steps (owner: compute + send, others: receive), so a caller that streams scenes
    spreads the NEXT scene's bank over the chunks of the current one and the transfers ride under the denoise kernels (RCCL runs on
    its own stream); `drain()` completes what is in flight except the newest `keep` messages; `finish()` returns the RefBank.
    All ranks must call advance() with the same step counts in the same order relative to the other collectives of `group`.
    `layers` (the per-step layout, identical for every step and scene) travels once: pass the previous stream's `.layers`."""

    def __init__(self, pipe, src: int, world_size: int, rank: int, device, n_steps: int, group=None, layers=None):
        from .sd.unet import RefBank
        self.pipe, self.src, self.world, self.rank, self.device, self.n = pipe, src, world_size, rank, device, n_steps
        self.group, self.layers = group, layers
        self.owner = rank == src
        self.i = 0                    # DDIM steps posted so far
        self.pending = []             # (step, work handle, flat buffer)
        self.tr = None
        self.bank = None if self.owner else RefBank()
        self._host_sync = None

    def begin(self, ref_z0, ref_disp, ctx_neg, ctx_pos):
        """owner only: start the reference trajectory (no-op elsewhere)"""
        if self.owner:
            self.tr = self.pipe.begin_ref_bank(ref_z0, ref_disp, ctx_neg, ctx_pos)
            self.bank = self.tr["bank"]
        return self

    def _needs_host_sync(self):
        if self._host_sync is None:
            import torch.distributed as dist
            self._host_sync = dist.get_backend(self.group) != "nccl"     # c10d's NCCL work orders itself after the current stream
        return self._host_sync

    def advance(self, nsteps: int | None = None):
        import torch.distributed as dist
        end = self.n if nsteps is None else min(self.n, self.i + nsteps)
        while self.i < end:
            i = self.i
            if self.owner:
                self.pipe.advance_ref_bank(self.tr, 1)
                if self.layers is None:
                    self.layers = _step_meta(self.bank, 0)
                    dist.broadcast_object_list([self.layers], src=self.src, group=self.group)
                flat = _pack_step(self.bank, i, self.layers)
                if flat.is_cuda and self._needs_host_sync():
                    torch.cuda.current_stream().synchronize()      # gloo reads the buffer from the host side
            else:
                if self.layers is None:
                    box = [None]
                    dist.broadcast_object_list(box, src=self.src, group=self.group)
                    self.layers = box[0]
                dtype = getattr(torch, self.layers[0][4].split(".")[-1])
                flat = torch.empty(_step_numel(self.layers), dtype=dtype, device=self.device)
            self.pending.append((i, dist.broadcast(flat, src=self.src, group=self.group, async_op=True), flat))
            self.i = i + 1
            self.drain(keep=MAX_INFLIGHT if nsteps is None else None)
        return self.i >= self.n

    def drain(self, keep: int | None = 0):
        """complete (and on the receivers unpack) everything in flight except the newest `keep` messages (None: nothing)"""
        if keep is None:
            return
        while len(self.pending) > keep:
            i, h, flat = self.pending.pop(0)
            h.wait()
            if not self.owner:
                _unpack_step(self.bank, i, self.layers, flat)

    def finish(self):
        assert self.i >= self.n, "advance() the stream to its last DDIM step first"
        self.drain(0)
        self.bank.mode = "use"
        return self.bank


def broadcast_ref_bank_pipelined(pipe, ref_z0, ref_disp, ctx_neg, ctx_pos, src: int, world_size: int, rank: int, device, n_steps: int,
                                 group=None):
    """The whole bank of one scene through a RefBankStream: the owner's step i+1 computes while step i travels; at most MAX_INFLIGHT
    packed / receive buffers exist beside the bank."""
    st = RefBankStream(pipe, src, world_size, rank, device, n_steps, group=group).begin(ref_z0, ref_disp, ctx_neg, ctx_pos)
    st.advance(None)
    return st.finish()


class FlatGrads:
    """The six leaf-gradient tensors as views of ONE flat fp32 allocation (what the fused backward's `grad_into` writes), so the
    gradient reduction of a training batch is a single RCCL all-reduce of that buffer with no gather copy (SURVEY.md 8e collective 2).
    `reduce_async()` posts it (c10d orders it after the work queued on the current stream); `wait()` makes the current stream wait for
    it -- call it before the buffer is read (optimizer) or overwritten (the next-but-one batch when two FlatGrads alternate)."""

    def __init__(self, params: dict, group=None, pad_to: int = 1):
        n = sum(int(v.numel()) for v in params.values())
        any_p = next(iter(params.values()))
        self.n = n
        self.flat = torch.zeros((n + pad_to - 1) // pad_to * pad_to, dtype=torch.float32, device=any_p.device)    # (pad_to: equal shards for ShardedAdam)
        self.views, o = {}, 0
        for k, v in params.items():
            self.views[k] = self.flat[o:o + v.numel()].view(v.shape)
            o += v.numel()
        self.group = group
        self.work = None

    def reduce_async(self, world_size: int):
        if world_size > 1:
            import torch.distributed as dist
            self.work = dist.all_reduce(self.flat, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None


class FlatParams:
    """The leaf parameters re-homed as views of ONE flat fp32 buffer with FlatGrads' layout (same order, same offsets), padded so that `world`
    equal shards of a multiple of 4 elements tile it: what ShardedAdam updates slice by slice and all-gathers in place.  The nn.Parameter objects
    stay the same (their `.data` becomes the view), so modules, optimizers' references and checkpoints are untouched."""

    def __init__(self, params: dict, world: int):
        n = sum(int(v.numel()) for v in params.values())
        any_p = next(iter(params.values()))
        self.n = n
        self.shard = ((n + world - 1) // world + 3) // 4 * 4
        self.flat = torch.zeros(self.shard * world, dtype=torch.float32, device=any_p.device)
        self.spans, o = {}, 0
        with torch.no_grad():
            for k, v in params.items():
                view = self.flat[o:o + v.numel()].view(v.shape)
                view.copy_(v.data)
                v.data = view
                self.spans[k] = (o, o + v.numel())
                o += v.numel()

    def matches(self, params: dict) -> bool:
        return all(k in self.spans and self.spans[k][1] - self.spans[k][0] == v.numel() and v.data_ptr() == self.flat.data_ptr() + 4 * self.spans[k][0]
                   for k, v in params.items())


def _hip_adam(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """torch.optim.Adam's update on one contiguous fp32 slice through the fused HIP kernel (gc_adam_step); raises on CPU tensors -- no fallback."""
    from . import _lib as L
    if not param.is_cuda:
        raise L.GaussCtrlHipError("ShardedAdam: the fused Adam kernel needs GPU tensors")
    L.check(L.lib().gc_adam_step(L.ptr(param), L.ptr(grad), L.ptr(exp_avg), L.ptr(exp_avg_sq), L.i64(param.numel()), L.f32(lr), L.f32(beta1), L.f32(beta2),
                                 L.f32(eps), L.i32(step), L.stream_ptr()), "gc_adam_step")


class ShardedAdam:
    """SURVEY.md 8e collective 2, second form: reduce-scatter of the flat gradient buffer -> every rank runs Adam on ITS 1 / world slice of the flat
    parameter buffer (the two moment buffers exist only for that slice: optimizer state 1 / world per rank) -> all-gather of the updated slices in
    place.  Same result as an all-reduce followed by the replicated per-group Adam (torch.optim.Adam semantics, /root/reference/gaussctrl/
    gc_config.py:58-87: per-group lr / eps), element for element: Adam is element-wise, and the reduce-scatter sums the same values.  Per step
    2 (G - 1) / G x 236 B x N cross the links either way; what changes is 8 bytes of moments and ~28 bytes of optimizer traffic per element / G.
    `adam` is the slice update (default: the fused HIP kernel); the gloo tests pass a reference implementation -- the product never does.
    Backends without reduce_scatter_tensor (gloo, the CPU / one-GPU tests) take all_reduce + slice."""

    def __init__(self, fparams: FlatParams, fgrads: FlatGrads, world: int, rank: int, group=None, betas=(0.9, 0.999), adam=None):
        assert fgrads.flat.numel() == fparams.flat.numel(), "FlatGrads must be built with pad_to = world * FlatParams.shard"
        self.fp, self.fg, self.world, self.rank, self.group, self.betas = fparams, fgrads, world, rank, group, betas
        self.lo, self.hi = rank * fparams.shard, (rank + 1) * fparams.shard
        self.exp_avg = torch.zeros(fparams.shard, dtype=torch.float32, device=fparams.flat.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.gshard = torch.zeros_like(self.exp_avg)
        self.steps = 0
        self.adam = adam or _hip_adam

    # ---- optimizer state that survives a change of the parameter set (culling) and a checkpoint
    def _gather_full(self, shard: torch.Tensor) -> torch.Tensor:
        """this rank's 1 / world slice of a moment buffer -> the whole flat buffer (FlatParams layout) on every rank"""
        if self.world <= 1:
            return shard.clone()
        import torch.distributed as dist
        out = torch.empty(self.world * shard.numel(), dtype=shard.dtype, device=shard.device)
        if shard.is_cuda and dist.get_backend(self.group) != "nccl":
            torch.cuda.current_stream().synchronize()
        dist.all_gather_into_tensor(out, shard.contiguous(), group=self.group)
        return out

    @torch.no_grad()
    def adopt(self, old: "ShardedAdam", keep: torch.Tensor) -> None:
        """Carry the Adam state of `old` (the optimizer of the parameter set BEFORE a cull) over to this one: `keep` is the boolean row mask
        the cull applied to every leaf tensor (gc_trainer.CullCallback / SplatfactoModel.cull_gaussians prune parameters and moments with the
        same mask in the replicated path).  The old moment shards are all-gathered, masked per tensor and re-sliced to the new shard
        boundaries; the step count (bias correction) continues.  A collective when world > 1: every rank culls at the same step."""
        n_old = int(keep.numel())
        ea, es = old._gather_full(old.exp_avg), old._gather_full(old.exp_avg_sq)
        new_a = torch.zeros(self.fp.flat.numel(), dtype=torch.float32, device=ea.device)
        new_s = torch.zeros_like(new_a)
        keep = keep.to(ea.device)
        for name, (a, b) in old.fp.spans.items():
            a2, b2 = self.fp.spans[name]
            w = (b - a) // n_old
            assert w * n_old == b - a and (b2 - a2) % w == 0, (name, a, b, a2, b2, n_old)
            new_a[a2:b2] = ea[a:b].view(n_old, w)[keep].reshape(-1)
            new_s[a2:b2] = es[a:b].view(n_old, w)[keep].reshape(-1)
        self.exp_avg.copy_(new_a[self.lo:self.hi])
        self.exp_avg_sq.copy_(new_s[self.lo:self.hi])
        self.steps = old.steps

    def state_dict(self, full: bool = True) -> dict:
        """full=True (a collective when world > 1): the moments of ALL parameters in FlatParams order, so that a checkpoint written at one world
        size restores at another; full=False: this rank's slice only."""
        if full:
            n = self.fp.n
            return {"layout": {k: list(v) for k, v in self.fp.spans.items()}, "steps": self.steps, "full": True,
                    "exp_avg": self._gather_full(self.exp_avg)[:n].clone(), "exp_avg_sq": self._gather_full(self.exp_avg_sq)[:n].clone()}
        return {"layout": {k: list(v) for k, v in self.fp.spans.items()}, "steps": self.steps, "full": False, "rank": self.rank, "world": self.world,
                "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone()}

    @torch.no_grad()
    def load_state_dict(self, sd: dict) -> None:
        if {k: list(v) for k, v in self.fp.spans.items()} != {k: list(v) for k, v in sd["layout"].items()}:
            raise ValueError("ShardedAdam.load_state_dict: the checkpoint's parameter layout differs from this model's")
        if sd.get("full", True):
            pad = self.fp.flat.numel() - self.fp.n
            for name, dst in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                src = torch.cat([sd[name].to(dst.device, torch.float32), torch.zeros(pad, device=dst.device)])
                dst.copy_(src[self.lo:self.hi])
        else:
            if sd["rank"] != self.rank or sd["world"] != self.world:
                raise ValueError("a per-rank ShardedAdam state restores only on the same rank / world size (save with full=True otherwise)")
            self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.steps = int(sd["steps"])

    @torch.no_grad()
    def step(self, hyper: dict):
        """hyper: parameter name -> (lr, eps) of this step (the trainer's schedulers own the learning rates)."""
        import torch.distributed as dist
        fp = self.fp
        if self.world > 1:
            if dist.get_backend(self.group) == "nccl":
                dist.reduce_scatter_tensor(self.gshard, self.fg.flat, group=self.group)
            else:
                if self.fg.flat.is_cuda:
                    torch.cuda.current_stream().synchronize()
                dist.all_reduce(self.fg.flat, group=self.group)
                self.gshard.copy_(self.fg.flat[self.lo:self.hi])
        else:
            self.gshard.copy_(self.fg.flat[self.lo:self.hi])
        self.steps += 1
        for name, (a, b) in fp.spans.items():
            x, y = max(a, self.lo), min(b, self.hi)
            if x >= y:
                continue
            lr, eps = hyper[name]
            self.adam(fp.flat[x:y], self.gshard[x - self.lo:y - self.lo], self.exp_avg[x - self.lo:y - self.lo], self.exp_avg_sq[x - self.lo:y - self.lo],
                      float(lr), self.betas[0], self.betas[1], float(eps), self.steps)
        if self.world > 1:
            if dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(fp.flat, fp.flat[self.lo:self.hi], group=self.group)          # in place: the input is this rank's slice of the output
            else:
                if fp.flat.is_cuda:
                    torch.cuda.current_stream().synchronize()
                dist.all_gather_into_tensor(fp.flat, fp.flat[self.lo:self.hi].clone(), group=self.group)
