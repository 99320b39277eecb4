"""Red-zone hunt for out-of-bounds WRITES: torch.empty / empty_like / zeros hand out views into buffers with 4 KB guard bands of 0xA5 on
both sides; after the small pipeline of tests/test_dist_gpu.py has run, every guard is checked and the creation site of each trampled
buffer is printed.  python scripts/oob_hunt.py [invariant] [ranks2 [allgather|owner]]"""
import math, os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
G = 4096
REG = []


def install():
    _empty, _zeros, _empty_like = torch.empty, torch.zeros, torch.empty_like

    def guarded(shape, dtype, device, zero):
        n = int(math.prod(shape)) if len(shape) else 1
        es = torch.empty((), dtype=dtype).element_size()
        nb = (n * es + 255) // 256 * 256
        raw = _empty(nb + 2 * G, dtype=torch.uint8, device=device)
        raw[:G].fill_(0xA5); raw[G + nb:].fill_(0xA5)
        if nb > n * es:
            raw[G + n * es:G + nb].fill_(0xA5)
        body = raw[G:G + n * es].view(dtype).view(*shape) if n else _empty(shape, dtype=dtype, device=device)
        if zero and n:
            body.zero_()
        site = "".join(traceback.format_stack(limit=6)[:-2]).replace(ROOT + "/", "")
        REG.append((raw, n * es, nb, site, tuple(shape), dtype))
        return body

    def on_gpu(device):
        return device is not None and torch.device(device).type == "cuda"

    def empty(*size, dtype=None, device=None, **k):
        if not on_gpu(device) or k.get("pin_memory") or k.get("memory_format") is not None:
            return _empty(*size, dtype=dtype, device=device, **k)
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        return guarded(shape, dtype or torch.float32, device, False)

    def zeros(*size, dtype=None, device=None, **k):
        if not on_gpu(device) or k:
            return _zeros(*size, dtype=dtype, device=device, **k)
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        return guarded(shape, dtype or torch.float32, device, True)

    def empty_like(t, **k):
        if not t.is_cuda or k or not t.is_contiguous():
            return _empty_like(t, **k)
        return guarded(tuple(t.shape), t.dtype, t.device, False)

    torch.empty, torch.zeros, torch.empty_like = empty, zeros, empty_like


def check(tag=""):
    torch.cuda.synchronize()
    bad = 0
    for raw, nbytes, nb, site, shape, dtype in REG:
        lo = raw[:G] != 0xA5
        hi = raw[G + nbytes:] != 0xA5
        if bool(lo.any()) or bool(hi.any()):
            bad += 1
            if bad <= 8:
                first_hi = int(hi.nonzero()[0]) if bool(hi.any()) else -1
                print(f"{tag}TRAMPLED guard of a {shape} {dtype} buffer: {int(lo.sum())} bytes before, {int(hi.sum())} bytes after (first at +{first_hi} past the end)\n{site}", flush=True)
    print(f"{tag}{len(REG)} guarded buffers, {bad} with a trampled guard", flush=True)


def _rank(rank, world, port, mode):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    install()
    from gaussctrl_amd.sd import ops
    ops.configure(ops.options_from_env())
    import test_dist_gpu as T
    pipe, model = T._build(world, rank, 0 if mode == "owner" else -1, mode == "allgather")
    T._run(pipe, model)
    check(f"[rank {rank}] ")
    dist.destroy_process_group()


if __name__ == "__main__":
    if "invariant" in sys.argv:
        os.environ["GC_BATCH_INVARIANT"] = "1"
    if "ranks2" in sys.argv:
        import torch.multiprocessing as mp
        mp.spawn(_rank, args=(2, 29787, "allgather" if "allgather" in sys.argv else "owner"), nprocs=2, join=True)
    else:
        install()
        from gaussctrl_amd.sd import ops
        ops.configure(ops.options_from_env())
        import test_dist_gpu as T
        pipe, model = T._build(1, 0, -1)
        T._run(pipe, model)
        check()
