"""Seeded inputs / weights of the production-geometry cross-view attention goldens (tests/golden/xview_big_*.npz).
The same recipe as tests/golden/make_xview_golden.py::big_inputs (which produced the reference outputs by importing the
reference's utils.py in the build container); CPU torch generators are deterministic across machines."""
import torch


def big_inputs(seed, f, L, H, D, Lt, Ct):
    g = torch.Generator().manual_seed(seed)
    C, B = H * D, 2 * f
    r = lambda *s: torch.randn(*s, generator=g)
    out = {}
    for kind, cin in (("self", C), ("text", Ct)):
        out[kind] = dict(x=r(B, L, C), ctx=None if kind == "self" else r(B, Lt, Ct),
                         wq=r(C, C) * (2.0 * C ** -0.5), wk=r(C, cin) * (2.0 * cin ** -0.5), wv=r(C, cin) * cin ** -0.5,
                         wo=r(C, C) * C ** -0.5, bo=r(C) * 0.1)
    return out
