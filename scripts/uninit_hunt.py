"""Hunt for reads of memory nobody wrote: torch.empty / empty_like are patched to hand out NaN-filled buffers and every ops.* call is
checked for NaN in its outputs (the first offender per op is printed).  `python scripts/uninit_hunt.py [invariant] [ranks2 [allgather|owner]]`
runs the small pipeline of tests/test_dist_gpu.py on one rank, or on two spawned ranks (both on cuda:0 over gloo)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def install(tag=""):
    _empty, _empty_like = torch.empty, torch.empty_like

    def _poison(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype == torch.uint8:
                t.fill_(0xFF)
        return t

    torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
    from gaussctrl_amd.sd import ops
    seen = set()

    def wrap(name, fn):
        def f(*a, **k):
            out = fn(*a, **k)
            outs = out if isinstance(out, tuple) else (out,)
            for i, o in enumerate(outs):
                if isinstance(o, torch.Tensor) and o.is_floating_point() and o.numel() and name not in seen:
                    bad = torch.isnan(o)
                    if bool(bad.any()):
                        seen.add(name)
                        shapes = [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)]
                        print(f"{tag}NaN in output {i} of ops.{name}: {int(bad.sum())} of {o.numel()} (shape {tuple(o.shape)}; inputs {shapes}; kwargs {sorted(k)})", flush=True)
            return out
        return f

    for n in ("linear", "conv3x3", "groupnorm", "groupnorm_coef", "layernorm", "concat_add", "axpby", "attention", "transformer_tail", "transformer_head",
              "cast_f32", "depth_to_disparity", "mask_composite"):
        setattr(ops, n, wrap(n, getattr(ops, n)))
    _ddim = ops.cfg_ddim_step

    def ddim(eps, lat, xin, *a):
        r = _ddim(eps, lat, xin, *a)
        if "ddim" not in seen and (bool(torch.isnan(lat).any()) or bool(torch.isnan(eps[..., :4]).any())):
            seen.add("ddim"); print(f"{tag}NaN at cfg_ddim_step: eps {int(torch.isnan(eps[..., :4]).sum())} lat {int(torch.isnan(lat).sum())}", flush=True)
        return r
    ops.cfg_ddim_step = ddim
    return ops


def _rank(rank, world, port, mode):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops = install(f"[rank {rank}] ")
    ops.configure(ops.options_from_env())
    import test_dist_gpu as T
    pipe, model = T._build(world, rank, 0 if mode == "owner" else -1, mode == "allgather")
    imgs, losses, means = T._run(pipe, model)
    print(f"[rank {rank}] nan in images {int(torch.isnan(imgs).sum())}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if "invariant" in sys.argv:
        os.environ["GC_BATCH_INVARIANT"] = "1"
    if "ranks2" in sys.argv:
        import torch.multiprocessing as mp
        mode = "allgather" if "allgather" in sys.argv else "owner"
        mp.spawn(_rank, args=(2, 29777, mode), nprocs=2, join=True)
    else:
        ops = install()
        ops.configure(ops.options_from_env())
        import test_dist_gpu as T
        pipe, model = T._build(1, 0, -1)
        imgs, losses, means = T._run(pipe, model)
        print("nan in images", int(torch.isnan(imgs).sum()))
