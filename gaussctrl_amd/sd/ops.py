"""Thin ctypes wrappers over the Part-B entry points of include/gaussctrl_hip.h.

Tensors are torch GPU tensors used only as device memory (2-byte activations in channels-last
"tokens x channels" layout, fp32 parameters for norms / biases); every arithmetic op is a hand-written HIP
kernel behind the C ABI.  No CPU / eager fallback exists.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os

import torch

from .. import _lib as L


class GemmDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("mode", C.c_int), ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("A", C.c_void_p), ("lda", C.c_int64),
                ("B", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("Cin", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("stride", C.c_int), ("upsample", C.c_int), ("pad_lo", C.c_int),
                ("W", C.c_void_p), ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("ld_rowvec", C.c_int64),
                ("rows_per_batch", C.c_int64), ("residual", C.c_void_p), ("ldr", C.c_int64), ("out_scale", C.c_float),
                ("act", C.c_int), ("geglu", C.c_int), ("out", C.c_void_p), ("ldc", C.c_int64), ("out_f32", C.c_int),
                ("out_t", C.c_void_p), ("ldt", C.c_int64), ("t_batch_stride", C.c_int64), ("t_col0", C.c_int64),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("zeros", C.c_void_p),
                ("ln_row_stats", C.c_void_p), ("ln_row_stat_slots", C.c_int), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float),
                ("out_row_stats", C.c_void_p), ("out_group_stats", C.c_void_p), ("gn_groups", C.c_int),
                ("fp8", C.c_int), ("w_scale", C.c_void_p), ("a_scale", C.c_int),
                ("kernel_variant", C.c_int), ("out_chan_parts", C.c_void_p), ("plan_rows", C.c_int64), ("out_fp8", C.c_int),
                ("w_set_rows", C.c_int64), ("w_set_stride", C.c_int64), ("softmax_keys", C.c_int)]


class AttnDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("batch", C.c_int), ("heads", C.c_int), ("head_dim", C.c_int), ("Lq", C.c_int),
                ("Lk", C.c_int), ("frames_per_half", C.c_int), ("nsets", C.c_int), ("set_kind", C.c_int * 5),
                ("set_weight", C.c_float * 5), ("scale", C.c_float),
                ("Q", C.c_void_p), ("ldq", C.c_int64), ("q_batch_stride", C.c_int64),
                ("K", C.c_void_p), ("ldk", C.c_int64), ("k_batch_stride", C.c_int64),
                ("Vt", C.c_void_p), ("ldvt", C.c_int64), ("vt_batch_stride", C.c_int64),
                ("O", C.c_void_p), ("ldo", C.c_int64), ("o_batch_stride", C.c_int64),
                ("Kref", C.c_void_p), ("kref_batch_stride", C.c_int64), ("Vtref", C.c_void_p),
                ("vtref_batch_stride", C.c_int64), ("ref_frames_per_half", C.c_int), ("q_prescaled", C.c_int),
                ("kernel_variant", C.c_int), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


@dataclasses.dataclass
class KernelOptions:
    """Every switch of the denoise host layer in ONE object (the C library holds no such state: kernel-selection overrides travel in the
    descriptors, gc_gemm_desc.kernel_variant / gc_attn_desc.kernel_variant / plan_rows).  The product default is `KernelOptions()`;
    GaussCtrlPipelineConfig.kernel_options / bench.py flags / tests pass other values through `configure(...)`.  Nothing on the product
    path reads the process environment: `options_from_env()` is the bridge the benchmark and experiment scripts call explicitly."""
    gemm_variant: int = 0            # gc_gemm_desc.kernel_variant (tile-height / kernel-family overrides, A/B and tests)
    attn_variant: int = 0            # gc_attn_desc.kernel_variant
    # Batch-invariant kernel planning (gc_gemm_desc.plan_rows, GroupNorm act bit 8): every output row is accumulated in an order that
    # depends on the layer's shape only, so a view's latents are bit-identical whatever shares its chunk or however many ranks shard
    # the scene (SURVEY.md 8e "bit-compatible with the single-GPU result").  Costs throughput (k-slices sized for one frame).
    batch_invariant: bool = False
    fused_head: bool = True          # level-0 transformer blocks: GroupNorm apply .. Q | K | V^T in one launch (csrc/dn_thead.hip)
    fused_tail: bool = True          # ... and everything after the self-attention in one launch (csrc/dn_ttail.hip)
    two_streams: bool = True         # ControlNet || UNet encoder on two HIP streams inside a denoise step
    gn_parts: bool = True            # GroupNorm statistics from the producer's epilogue (per-channel partials, plain stores): the
                                     # stand-alone statistics + finalize launches disappear (1 launch per GroupNorm instead of 3)
    tail_in_rows: bool = True        # CFG-shared prefix: the level-0 tail kernel reads the shared rows for both halves (gc_ttail_desc.in_rows); False: 3 duplicate copies
    q_only: bool = True              # ControlNet blocks against a cached bank (self weight 0): project Q only, not Q | K | V (GC_Q_ONLY=0)
    ffout_merge: bool = True         # LayerNorm-folded blocks: feed-forward down projection + proj_out as ONE GEMM over [ff | h] (GC_FFOUT_MERGE=0)
    text_fold: bool = True           # LayerNorm-folded blocks: attn2.to_q -> text attention -> attn2.to_out as two GEMMs (SDNet._text_fold; GC_TEXT_FOLD=0)
    cfg_share: bool = True           # CFG-shared prefix (sd.unet.AttnCtx.share): conv_in .. the first transformer block's self-attention computed for
                                     # ONE of the two identical CFG halves (bench: GC_CFG_SHARE=0 restores the duplicated computation)
    fp8_min_hw: int = 256            # fp8 path (weights.add_fp8_convs): smallest map, in pixels, whose resnet convolutions run on e4m3 -- 16 x 16 maps
                                     # run k-sliced; the 8 x 8 maps measure slower in e4m3 than on the split-K bf16 kernels
    ablate: frozenset = frozenset()  # TIMING ablations (results wrong by construction; scripts/ablate_classes.sh): op classes whose
                                     # launches are skipped -- {"gn", "ln", "attn", "linear", "conv", "head", "tail"}


OPTIONS = KernelOptions()


def configure(options: KernelOptions | None = None, **kw) -> KernelOptions:
    """Replace (options=) or update (**kw) the process-wide kernel options; returns the active object."""
    global OPTIONS, BATCH_INVARIANT
    if options is not None:
        OPTIONS = options
    if kw:
        OPTIONS = dataclasses.replace(OPTIONS, **kw)
    BATCH_INVARIANT = OPTIONS.batch_invariant
    KERNEL_VARIANT["gemm"], KERNEL_VARIANT["attn"] = OPTIONS.gemm_variant, OPTIONS.attn_variant
    return OPTIONS


def options_from_env(env=None) -> KernelOptions:
    """The experiment switches of bench.py / scripts / the spawned ranks of the tests, decoded from GC_* variables (NOT called on
    import and never by the plugin path):  GC_GEMM_MT / GC_GEMM8 / GC_GEMM_CONVSPLIT / GC_GEMM_DBG / GC_GEMM_PW -> gemm_variant;  GC_ATTN_SAFE /
    GC_ATTN_16 / GC_ATTN_V -> attn_variant;  GC_BATCH_INVARIANT, GC_FUSED_HEAD, GC_FUSED_TAIL, GC_DN_STREAMS, GC_GN_PARTS, GC_ABLATE=a,b,c."""
    e = os.environ if env is None else env
    on = lambda k, d: e.get(k, d) not in ("", "0")
    g = int(e.get("GC_GEMM_MT", "0")) & 7
    use8 = e.get("GC_GEMM8")
    if use8 == "0":
        g |= 0x10
    elif use8 == "2":
        g |= 0x20
    cs = e.get("GC_GEMM_CONVSPLIT")
    if cs == "0":
        g |= 0x40
    elif cs == "1":            # (2 = default: also the small grids -- 8 x 8 maps, stride-2 convs -- on the k-sliced 8-wave kernel)
        g |= 0x80
    g |= (int(e.get("GC_GEMM_DBG", "0")) & 0xff) << 8
    g |= (int(e.get("GC_GEMM_SPLIT_MT", "0")) & 7) << 24                  # m-tiles per wave of the k-sliced 8-wave problems (default 2)
    if e.get("GC_GEMM_PW", "") != "":            # forced column-panel width of the GEMM tile order (0 = whole rows, the round-1..4 order)
        g |= ((int(e["GC_GEMM_PW"]) + 1) & 0xff) << 16
    a = (1 if on("GC_ATTN_SAFE", "0") else 0) | (2 if on("GC_ATTN_16", "0") else 0) | (int(e.get("GC_ATTN_V", "0")) << 2)
    return KernelOptions(gemm_variant=g, attn_variant=a, batch_invariant=on("GC_BATCH_INVARIANT", "0"),
                         fused_head=on("GC_FUSED_HEAD", "1"), fused_tail=on("GC_FUSED_TAIL", "1"), two_streams=on("GC_DN_STREAMS", "1"),
                         gn_parts=on("GC_GN_PARTS", "1"), cfg_share=on("GC_CFG_SHARE", "1"), text_fold=on("GC_TEXT_FOLD", "1"), ffout_merge=on("GC_FFOUT_MERGE", "1"), q_only=on("GC_Q_ONLY", "1"), tail_in_rows=on("GC_TAIL_INROWS", "1"),
                         ablate=frozenset(x for x in e.get("GC_ABLATE", "").split(",") if x))


KERNEL_VARIANT = {"gemm": 0, "attn": 0}     # mirrors of OPTIONS kept as module attributes (read on every launch; tests patch them)
BATCH_INVARIANT = False


DT = {torch.bfloat16: 0, torch.float16: 1}
_abl_cache = {}


def _ablated(cls, shape, dtype, device):
    """TIMING ablation (KernelOptions.ablate): the op class is skipped and a cached all-zero tensor of the output's shape is handed on
    (results wrong by construction; zeros keep every downstream kernel on its normal code path).  None when the class is live."""
    if cls not in OPTIONS.ablate:
        return None
    key = (cls, tuple(shape), dtype, device, stream_handle())
    t = _abl_cache.get(key)
    if t is None:
        t = _abl_cache[key] = torch.zeros(tuple(shape), dtype=dtype, device=device)
    return t


def _dt(t):
    try:
        return DT[t.dtype]
    except KeyError:
        raise L.GaussCtrlHipError(f"denoise kernels take bf16 / f16 activations, got {t.dtype}")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.GaussCtrlHipError("denoise ops need GPU tensors (HIP path only; no CPU fallback)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_handle():
    """raw hipStream_t of torch's current stream on the current device.  The host enqueues ~13 000 launches per 3-view chunk: through
    torch.cuda.current_stream() this lookup alone was 84 of the 275 ms of host time per chunk (scripts/cpu_bound_check.py)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return C.c_void_p(stream_handle())


_zero_page = {}


class ChanParts:
    """Partial (sum, sum^2) of a tensor per row slab and GroupNorm group, left by its PRODUCER (gc_gemm_desc.out_chan_parts /
    gc_dn_concat_add_parts) for the GroupNorm that follows (groupnorm(..., parts=)): buf fp32 [B, nslab, G, 2 halves, 2]; rows = rows per
    slab; mode 0: slabs are the producer's row tiles counted over all rows, 1: slabs restart at every batch; col_tile = the producer's
    column tile (half 1 holds the rest of a group that straddles two of them); groups = G."""
    __slots__ = ("buf", "rows", "nslab", "mode", "col_tile", "groups")

    def __init__(self, buf, rows, nslab, mode, col_tile, groups):
        self.buf, self.rows, self.nslab, self.mode, self.col_tile, self.groups = buf, rows, nslab, mode, col_tile, groups


class RowStats:
    """(sum, sum^2) of every output row of a GEMM, left by its epilogue as [slots][M][2] partials over column slabs (plain stores: no
    zero-init); handed to the GEMM that has the following LayerNorm folded in (linear(..., ln=(row_stats, colsum, eps)))."""
    __slots__ = ("buf", "slots")

    def __init__(self):
        self.buf, self.slots = None, 0


DEBUG_FILL = None      # tests only: value the statistics buffers (RowStats / ChanParts) are filled with BEFORE the producer runs -- they come from
                       # torch.empty and the consumers trust the producers' slot layout: NaN here exposes any slot read but never written


def _run_gemm(d, dev, what, row_stats=None, want_parts=False, gn_groups=32):
    """-> ChanParts of the output when want_parts and the kernel this problem selects can produce them, else None"""
    lib = L.lib()
    z = _zero_page.get(dev)
    if z is None:
        z = _zero_page[dev] = torch.zeros(64, dtype=torch.uint8, device=dev)
    d.zeros = z.data_ptr()
    d.kernel_variant = KERNEL_VARIANT["gemm"]
    wsb = lib.gc_dn_gemm_workspace_bytes(C.byref(d))
    ws = None
    if wsb:
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)       # caching allocator: no hipMalloc on the hot path
        d.workspace = ws.data_ptr(); d.workspace_bytes = wsb
    if row_stats is not None:                      # the slab count depends on the kernel the library picks for this problem
        row_stats.slots = int(lib.gc_dn_gemm_row_stat_slots(C.byref(d)))
        row_stats.buf = torch.empty(row_stats.slots, d.M, 2, dtype=torch.float32, device=dev)
        if DEBUG_FILL is not None:
            row_stats.buf.fill_(DEBUG_FILL)
        d.out_row_stats = row_stats.buf.data_ptr()
    parts = None
    if want_parts and d.rows_per_batch >= 256:
        rows, ns, ct = C.c_int64(0), C.c_int(0), C.c_int(0)
        d.gn_groups = gn_groups
        L.check(lib.gc_dn_gemm_chan_parts_layout(C.byref(d), C.byref(rows), C.byref(ns), C.byref(ct)), "gc_dn_gemm_chan_parts_layout")
        if 0 < rows.value and ns.value <= 64:        # (more slabs -- the VAE's 256 x 256 / 512 x 512 maps -- would make the apply prologue the long pole)
            parts = ChanParts(torch.empty(d.M // d.rows_per_batch, ns.value, gn_groups, 2, 2, dtype=torch.float32, device=dev), rows.value, ns.value,
                              0, ct.value, gn_groups)
            if DEBUG_FILL is not None:
                parts.buf.fill_(DEBUG_FILL)
            d.out_chan_parts = parts.buf.data_ptr()
        else:
            d.gn_groups = 0
    L.check(lib.gc_dn_gemm(C.byref(d), _stream()), what)
    return parts


def _stats_args(d, ln, group_stats):
    if ln is not None:                          # (RowStats of x, column sums [N] of the gamma-folded weights, eps)
        d.ln_row_stats = ln[0].buf.data_ptr(); d.ln_row_stat_slots = ln[0].slots; d.ln_colsum = ln[1].data_ptr(); d.ln_eps = ln[2]
    if group_stats is not None:                 # zeroed fp32 [B, G, 2]
        d.out_group_stats = group_stats.data_ptr(); d.gn_groups = group_stats.shape[-2]


def linear(x, w, bias=None, residual=None, act=0, geglu=False, out_f32=False, scale=1.0, rowvec=None, rows_per_batch=0,
           ld_rowvec=None, out=None, want_out=True, out_t=None, ldt=0, t_batch_stride=0, t_col0=0, out_cols=None,
           ln=None, row_stats=None, group_stats=None, chan_parts=False, gn_groups=32, w_set_rows=0, softmax_keys=0):
    """x [..., K] (last dim contiguous, rows strided by x.stride(-2)) @ w[N,K]^T with fused epilogue.
    w_set_rows > 0: w is [S, N, K] (bias / colsum [S, N]): rows [s w_set_rows, (s + 1) w_set_rows) use matrix s (lean LayerNorm-fold problems only);
    softmax_keys > 0 (with ln): every 80-column block of a row is one head's scores, the output is their softmax over the first softmax_keys columns.
    chan_parts=True: returns (out, ChanParts | None) -- the per-channel partial sums of the output for a following GroupNorm (needs rows_per_batch);
    ln=(RowStats of x, colsum [N], eps): LayerNorm folded in (w carries gamma, bias carries W beta);
    row_stats: a RowStats() that receives the per-row (sum, sum^2) partials of the stored output;
    group_stats: zeroed fp32 [B, G, 2] that receives the per-(batch, GroupNorm group) sums (needs rows_per_batch)."""
    _gpu(x, w)
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[-2]
    lda = x.stride(-2) if x.dim() > 1 else K
    d = GemmDesc()
    d.dtype = _dt(x); d.mode = 0; d.M, d.N, d.K = M, N, K
    d.A = x.data_ptr(); d.lda = lda; d.W = w.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    if w_set_rows:
        assert w.dim() == 3 and w.is_contiguous() and (bias is None or bias.shape == (w.shape[0], N)) and M == w.shape[0] * w_set_rows
        d.w_set_rows = int(w_set_rows); d.w_set_stride = N * K
    d.softmax_keys = int(softmax_keys)
    if rowvec is not None:
        d.rowvec = rowvec.data_ptr(); d.ld_rowvec = rowvec.stride(0) if ld_rowvec is None else ld_rowvec
    d.rows_per_batch = rows_per_batch
    if BATCH_INVARIANT and x.dim() >= 3:
        d.plan_rows = M // x.shape[0]                 # the rows one frame contributes
    No = N // 2 if geglu else (N if out_cols is None else out_cols)
    if OPTIONS.ablate and out is None and row_stats is None and M >= 256:
        z = _ablated("linear", x.shape[:-1] + (No,), torch.float32 if out_f32 else x.dtype, x.device)
        z = _ablated(f"linear_k{K}", x.shape[:-1] + (No,), torch.float32 if out_f32 else x.dtype, x.device) if z is None else z
        if z is not None:
            return ((z if want_out else None), None) if chan_parts else (z if want_out else None)
    if residual is not None:
        d.residual = residual.data_ptr(); d.ldr = residual.stride(-2)
    d.out_scale = scale; d.act = act; d.geglu = int(geglu)
    if want_out:
        if out is None:
            out = torch.empty(x.shape[:-1] + (No,), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
        d.out = out.data_ptr(); d.ldc = out.stride(-2); d.out_f32 = int(out_f32)
    if out_t is not None:
        d.out_t = out_t.data_ptr(); d.ldt = ldt; d.t_batch_stride = t_batch_stride; d.t_col0 = t_col0
    _stats_args(d, ln, group_stats)
    parts = _run_gemm(d, x.device, "gc_dn_gemm", row_stats, want_parts=chan_parts and not BATCH_INVARIANT, gn_groups=gn_groups)
    return (out, parts) if chan_parts else out


def conv3x3(x, w, bias=None, stride=1, upsample=False, rowvec=None, ld_rowvec=None, residual=None, act=0, scale=1.0,
            out_f32=False, pad_lo=1, group_stats=None, chan_parts=False, gn_groups=32):
    """x [B,H,W,Cin] NHWC, w [N, 9*Cin] ((tap, cin) order), pad 1.  chan_parts=True: returns (out, ChanParts | None)."""
    _gpu(x, w)
    B, H, W_, Cin = x.shape
    Hin, Win = (2 * H, 2 * W_) if upsample else (H, W_)
    # pad_lo=1: Conv2d(padding=1); pad_lo=0: F.pad(x,(0,1,0,1)) + Conv2d(padding=0) (VAE encoder downsample)
    Ho, Wo = (Hin + pad_lo + 1 - 3) // stride + 1, (Win + pad_lo + 1 - 3) // stride + 1
    N = w.shape[0]
    if OPTIONS.ablate:
        z = _ablated("conv", (B, Ho, Wo, N), torch.float32 if out_f32 else x.dtype, x.device)
        z = _ablated(f"conv_hw{Ho}", (B, Ho, Wo, N), torch.float32 if out_f32 else x.dtype, x.device) if z is None else z
        if z is not None:
            return (z, None) if chan_parts else z
    out = torch.empty(B, Ho, Wo, N, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    d = GemmDesc()
    d.dtype = _dt(x); d.mode = 1; d.M, d.N, d.K = B * Ho * Wo, N, 9 * Cin
    d.A = x.data_ptr(); d.lda = Cin
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.stride, d.upsample = B, H, W_, Cin, Ho, Wo, stride, int(upsample)
    d.pad_lo = pad_lo
    d.W = w.data_ptr(); d.bias = None if bias is None else bias.data_ptr()
    if rowvec is not None:
        d.rowvec = rowvec.data_ptr(); d.ld_rowvec = rowvec.stride(0) if ld_rowvec is None else ld_rowvec
    d.rows_per_batch = Ho * Wo
    if BATCH_INVARIANT:
        d.plan_rows = Ho * Wo
    if residual is not None:
        d.residual = residual.data_ptr(); d.ldr = N
    d.out_scale = scale; d.act = act
    d.out = out.data_ptr(); d.ldc = N; d.out_f32 = int(out_f32)
    _stats_args(d, None, group_stats)
    parts = _run_gemm(d, x.device, "gc_dn_gemm(conv3x3)", want_parts=chan_parts and not BATCH_INVARIANT, gn_groups=gn_groups)
    return (out, parts) if chan_parts else out


_gn_ws = {}


def groupnorm(x, gamma, beta, groups, eps, silu, parts=None):
    """x [B,H,W,C] (or [B,HW,C]).  parts: the ChanParts x's producer left -> ONE launch (gc_dn_groupnorm_apply_parts) instead of three."""
    _gpu(x)
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    if OPTIONS.ablate:
        z = _ablated("gn", x.shape, x.dtype, x.device)
        if z is not None:
            return z
    if parts is not None:
        assert parts.groups == groups
        y = torch.empty_like(x)
        L.check(L.lib().gc_dn_groupnorm_apply_parts(_dt(x), _p(x), _p(y), C.c_int64(B), C.c_int64(HW), Cc, groups, _p(gamma), _p(beta), C.c_float(eps),
                                                    int(silu), _p(parts.buf), C.c_int64(parts.rows), parts.nslab, parts.mode, parts.col_tile, _stream()),
                "gc_dn_groupnorm_apply_parts")
        return y
    key = (x.device, B, HW, Cc, stream_handle())      # scratch is per stream: two networks may run concurrently
    ws = _gn_ws.get(key)
    if ws is None:
        nbytes = L.lib().gc_dn_groupnorm_workspace_bytes(C.c_int64(B), C.c_int64(HW), Cc)
        ws = _gn_ws[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    L.check(L.lib().gc_dn_groupnorm(_dt(x), _p(x), _p(y), C.c_int64(B), C.c_int64(HW), Cc, groups, _p(gamma), _p(beta),
                                    C.c_float(eps), int(silu) | (0x100 if BATCH_INVARIANT else 0), _p(ws), _stream()), "gc_dn_groupnorm")
    return y


def groupnorm_apply(x, group_stats, gamma, beta, groups, eps, silu):
    """GroupNorm(+SiLU) of x [B,H,W,C] (or [B,HW,C]) from the per-(batch, group) sums [B, G, 2] its producer accumulated: ONE launch."""
    _gpu(x, group_stats)
    chan_stats = group_stats
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    y = torch.empty_like(x)
    L.check(L.lib().gc_dn_groupnorm_apply(_dt(x), _p(x), _p(y), C.c_int64(B), C.c_int64(HW), Cc, groups, _p(gamma), _p(beta),
                                          C.c_float(eps), int(silu), _p(chan_stats), _stream()), "gc_dn_groupnorm_apply")
    return y


def pad128(c):
    return (c + 127) // 128 * 128


def groupnorm_apply_fp8(x, group_stats, gamma, beta, groups, eps, silu, a_scale=127):
    """GroupNorm(+SiLU) with an e4m3 output [B,H,W,pad128(C)] (uint8) for conv3x3_fp8; a_scale = E8M0 byte of the tensor-wide scale."""
    _gpu(x, group_stats)
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    Cp = pad128(Cc)
    y = torch.empty(x.shape[:-1] + (Cp,), dtype=torch.uint8, device=x.device)
    L.check(L.lib().gc_dn_groupnorm_apply_fp8(_dt(x), _p(x), _p(y), C.c_int64(B), C.c_int64(HW), Cc, Cp, groups, _p(gamma), _p(beta),
                                              C.c_float(eps), int(silu), _p(group_stats), int(a_scale), _stream()), "gc_dn_groupnorm_apply_fp8")
    return y


def group_stats(x, gs):
    """per-(batch, GroupNorm group) (sum, sum^2) of x [B, ..., C], ADDED into the zeroed fp32 gs [B, G, 2] (one streaming launch)."""
    _gpu(x, gs)
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    L.check(L.lib().gc_dn_group_stats(_dt(x), _p(x), C.c_int64(B), C.c_int64(HW), Cc, gs.shape[-2], _p(gs), _stream()), "gc_dn_group_stats")
    return gs


def groupnorm_apply_parts_fp8(x, parts, gamma, beta, groups, eps, silu, a_scale=127):
    """GroupNorm(+SiLU) -> e4m3 [B,H,W,pad128(C)] in ONE launch from the ChanParts x's producer left (gc_dn_groupnorm_apply_parts_fp8)."""
    _gpu(x)
    Cc = x.shape[-1]
    B = x.shape[0]
    HW = x.numel() // (B * Cc)
    Cp = pad128(Cc)
    assert parts.groups == groups
    y = torch.empty(x.shape[:-1] + (Cp,), dtype=torch.uint8, device=x.device)
    L.check(L.lib().gc_dn_groupnorm_apply_parts_fp8(_dt(x), _p(x), _p(y), C.c_int64(B), C.c_int64(HW), Cc, Cp, groups, _p(gamma), _p(beta), C.c_float(eps),
                                                    int(silu), _p(parts.buf), C.c_int64(parts.rows), parts.nslab, parts.mode, parts.col_tile, int(a_scale),
                                                    _stream()), "gc_dn_groupnorm_apply_parts_fp8")
    return y


def groupnorm_fp8(x, gamma, beta, groups, eps, silu, a_scale=127):
    """Stand-alone GroupNorm(+SiLU) -> e4m3: one statistics launch (gc_dn_group_stats) + the quantising apply."""
    _gpu(x)
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    gs = torch.zeros(B, groups, 2, dtype=torch.float32, device=x.device)
    L.check(L.lib().gc_dn_group_stats(_dt(x), _p(x), C.c_int64(B), C.c_int64(HW), Cc, groups, _p(gs), _stream()), "gc_dn_group_stats")
    return groupnorm_apply_fp8(x, gs, gamma, beta, groups, eps, silu, a_scale)


def conv3x3_fp8(x8, w8, w_scale, out_dtype, bias=None, stride=1, rowvec=None, ld_rowvec=None, residual=None, act=0, scale=1.0, a_scale=127,
                group_stats=None, chan_parts=False, gn_groups=32):
    """3x3 conv (pad 1) on e4m3 operands with the block-scaled MFMA: x8 [B,H,W,Cp] uint8 (Cp % 128 == 0), w8 [N, 9*Cp] uint8 ((tap, cin)
    order), w_scale [N] uint8 E8M0 per output channel; output in `out_dtype` (bf16 / f16) with the usual fused epilogue."""
    _gpu(x8, w8, w_scale)
    B, H, W_, Cp = x8.shape
    Ho, Wo = (H + 2 - 3) // stride + 1, (W_ + 2 - 3) // stride + 1
    N = w8.shape[0]
    out = torch.empty(B, Ho, Wo, N, dtype=out_dtype, device=x8.device)
    d = GemmDesc()
    d.dtype = DT[out_dtype]; d.mode = 1; d.M, d.N, d.K = B * Ho * Wo, N, 9 * Cp
    d.A = x8.data_ptr(); d.lda = Cp
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.stride, d.upsample = B, H, W_, Cp, Ho, Wo, stride, 0
    d.pad_lo = 1
    d.W = w8.data_ptr(); d.bias = None if bias is None else bias.data_ptr()
    if rowvec is not None:
        d.rowvec = rowvec.data_ptr(); d.ld_rowvec = rowvec.stride(0) if ld_rowvec is None else ld_rowvec
    d.rows_per_batch = Ho * Wo
    if BATCH_INVARIANT:
        d.plan_rows = Ho * Wo
    if residual is not None:
        d.residual = residual.data_ptr(); d.ldr = N
    d.out_scale = scale; d.act = act
    d.out = out.data_ptr(); d.ldc = N; d.out_f32 = 0
    d.fp8 = 1; d.w_scale = w_scale.data_ptr(); d.a_scale = int(a_scale)
    _stats_args(d, None, group_stats)
    # chan_parts=True: returns (out, ChanParts | None) -- the k-sliced problems (16 x 16 maps) leave the GroupNorm partials of their output
    parts = _run_gemm(d, x8.device, "gc_dn_gemm(conv3x3 fp8)", want_parts=chan_parts and not BATCH_INVARIANT, gn_groups=gn_groups)
    return (out, parts) if chan_parts else out


def linear_fp8(x8, w8, w_scale, out_dtype, bias=None, residual=None, act=0, scale=1.0, a_scale=127, rows_per_batch=0, row_stats=None,
               group_stats=None, ln=None, geglu=False, out_fp8=0, out_t=None, ldt=0, t_batch_stride=0, t_col0=0, out_cols=None):
    """x8 [..., K] e4m3 bytes @ w8[N, K]^T (K % 128 == 0) on the block-scaled MFMA, output `out_dtype` -- or, with out_fp8 = an E8M0 byte,
    e4m3 bytes of value * 2^(127 - out_fp8) (uint8 tensor: the operand of the next fp8 linear).  geglu: rows of w8 / bias permuted by
    weights.geglu_permute, output [..., N / 2]; out_t / ldt / t_batch_stride / t_col0 / out_cols: the fused Q | K | V^T layout of `linear`."""
    _gpu(x8, w8, w_scale)
    K = x8.shape[-1]
    M = x8.numel() // K
    N = w8.shape[0]
    No = N // 2 if geglu else (N if out_cols is None else out_cols)
    out = torch.empty(x8.shape[:-1] + (No,), dtype=torch.uint8 if out_fp8 else out_dtype, device=x8.device)
    d = GemmDesc()
    d.dtype = DT[out_dtype]; d.mode = 0; d.M, d.N, d.K = M, N, K
    d.A = x8.data_ptr(); d.lda = x8.stride(-2) if x8.dim() > 1 else K; d.W = w8.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    d.rows_per_batch = rows_per_batch
    if BATCH_INVARIANT and x8.dim() >= 3:
        d.plan_rows = M // x8.shape[0]                # the rows one frame contributes
    if residual is not None:
        d.residual = residual.data_ptr(); d.ldr = residual.stride(-2)
    d.out_scale = scale; d.act = act; d.geglu = int(geglu)
    d.out = out.data_ptr(); d.ldc = out.stride(-2) if out.dim() > 1 else No; d.out_fp8 = int(out_fp8)
    if out_t is not None:
        d.out_t = out_t.data_ptr(); d.ldt = ldt; d.t_batch_stride = t_batch_stride; d.t_col0 = t_col0
    d.fp8 = 1; d.w_scale = w_scale.data_ptr(); d.a_scale = int(a_scale)
    _stats_args(d, ln, group_stats)
    _run_gemm(d, x8.device, "gc_dn_gemm(linear fp8)", row_stats)
    return out


def layernorm_fp8(x, gamma, beta, eps=1e-5, a_scale=127):
    """LayerNorm with an e4m3 output [.., C] (uint8) for linear_fp8; a_scale = E8M0 byte of the tensor-wide scale (stored = y * 2^(127 - a_scale))."""
    _gpu(x)
    Cc = x.shape[-1]
    y = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    L.check(L.lib().gc_dn_layernorm_fp8(_dt(x), _p(x), _p(y), C.c_int64(x.numel() // Cc), Cc, _p(gamma), _p(beta),
                                        C.c_float(eps), int(a_scale), _stream()), "gc_dn_layernorm_fp8")
    return y


def layernorm(x, gamma, beta, eps=1e-5):
    _gpu(x)
    Cc = x.shape[-1]
    if OPTIONS.ablate:
        z = _ablated("ln", x.shape, x.dtype, x.device)
        if z is not None:
            return z
    y = torch.empty_like(x)
    L.check(L.lib().gc_dn_layernorm(_dt(x), _p(x), _p(y), C.c_int64(x.numel() // Cc), Cc, _p(gamma), _p(beta),
                                    C.c_float(eps), _stream()), "gc_dn_layernorm")
    return y


def concat_add(a, b, c=None, group_stats=None, chan_parts=False, gn_groups=32):
    """cat([a, b (+ c)], dim=-1) on channels-last tensors [B, ..., C]; group_stats (zeroed fp32 [B, G, 2]) receives the
    per-(batch, GroupNorm group) (sum, sum^2) of the output; chan_parts=True: returns (out, ChanParts | None) instead."""
    _gpu(a, b, c)
    C1, C2 = a.shape[-1], b.shape[-1]
    M = a.numel() // C1
    out = torch.empty(a.shape[:-1] + (C1 + C2,), dtype=a.dtype, device=a.device)
    if chan_parts:
        rpb = M // a.shape[0]
        if rpb < 256 or BATCH_INVARIANT or "gn" in OPTIONS.ablate:
            return concat_add(a, b, c), None
        rows, ns, ct = C.c_int64(0), C.c_int(0), C.c_int(0)
        L.check(L.lib().gc_dn_concat_parts_layout(C.c_int64(rpb), C1 + C2, gn_groups, C.byref(rows), C.byref(ns), C.byref(ct)), "gc_dn_concat_parts_layout")
        if rows.value == 0:
            return concat_add(a, b, c), None
        parts = ChanParts(torch.empty(a.shape[0], ns.value, gn_groups, 2, 2, dtype=torch.float32, device=a.device), rows.value, ns.value, 1, ct.value, gn_groups)
        if DEBUG_FILL is not None:
            parts.buf.fill_(DEBUG_FILL)
        L.check(L.lib().gc_dn_concat_add_parts(_dt(a), _p(a), C1, _p(b), _p(c), C2, _p(out), C.c_int64(M), C.c_int64(rpb), gn_groups, _p(parts.buf), _stream()),
                "gc_dn_concat_add_parts")
        return out, parts
    L.check(L.lib().gc_dn_concat_add(_dt(a), _p(a), C1, _p(b), _p(c), C2, _p(out), C.c_int64(M), C.c_int64(M // a.shape[0]),
                                     _p(group_stats), 0 if group_stats is None else group_stats.shape[-2], _stream()), "gc_dn_concat_add")
    return out


def axpby(a, sa=1.0, b=None, sb=1.0, act=0):
    _gpu(a, b)
    out = torch.empty_like(a)
    L.check(L.lib().gc_dn_axpby(_dt(a), _p(a), C.c_float(sa), _p(b), C.c_float(sb), act, _p(out), C.c_int64(a.numel()),
                                _stream()), "gc_dn_axpby")
    return out


def cast_f32(a, dtype, silu=False):
    _gpu(a)
    out = torch.empty(a.shape, dtype=dtype, device=a.device)
    L.check(L.lib().gc_dn_cast_f32(DT[dtype], _p(a), int(silu), _p(out), C.c_int64(a.numel()), _stream()), "gc_dn_cast_f32")
    return out


def softmax_rows_(s, scale):
    _gpu(s)
    N = s.shape[-1]
    L.check(L.lib().gc_dn_softmax_rows(_dt(s), _p(s), C.c_int64(s.numel() // N), C.c_int64(N), C.c_int64(s.stride(-2)),
                                       C.c_float(scale), _stream()), "gc_dn_softmax_rows")
    return s


def attention(q, k, vt, heads, sets, frames_per_half, Lk=None, kref=None, vtref=None, ref_fph=0, scale=None, q_prescaled=False):
    """q [B,Lq,C], k [Bk,Lk,C], vt [Bk,C,Lkp] (token-contiguous, zero padded).  sets: [(kind, weight)]:
    kind -1 = own frame, -2 = frame b // frames_per_half, r >= 0 = reference r (bank = kref/vtref or k/vt)."""
    _gpu(q, k, vt)
    B, Lq, Cc = q.shape
    D = Cc // heads
    Lk = k.shape[1] if Lk is None else Lk
    if OPTIONS.ablate:
        z = _ablated("attn", (B, Lq, Cc), q.dtype, q.device)
        z = _ablated(f"attn{D}", (B, Lq, Cc), q.dtype, q.device) if z is None else z
        if z is not None:
            return z
    o = torch.empty(B, Lq, Cc, dtype=q.dtype, device=q.device)
    d = AttnDesc()
    d.dtype = _dt(q); d.batch, d.heads, d.head_dim, d.Lq, d.Lk = B, heads, D, Lq, Lk
    d.frames_per_half = frames_per_half; d.nsets = len(sets)
    for i, (kind, w) in enumerate(sets):
        d.set_kind[i] = kind; d.set_weight[i] = w
    d.scale = (D ** -0.5) if scale is None else scale
    d.q_prescaled = int(q_prescaled)     # Q carries scale*log2(e) already (weights.prepare(..., fold_attn_scale_heads=...))
    d.kernel_variant = KERNEL_VARIANT["attn"]
    d.Q = q.data_ptr(); d.ldq = q.stride(1); d.q_batch_stride = q.stride(0)
    d.K = k.data_ptr(); d.ldk = k.stride(1); d.k_batch_stride = k.stride(0)
    d.Vt = vt.data_ptr(); d.ldvt = vt.stride(1); d.vt_batch_stride = vt.stride(0)
    d.O = o.data_ptr(); d.ldo = Cc; d.o_batch_stride = Lq * Cc
    if kref is not None:
        d.Kref = kref.data_ptr(); d.kref_batch_stride = kref.stride(0)
        d.Vtref = vtref.data_ptr(); d.vtref_batch_stride = vtref.stride(0); d.ref_frames_per_half = ref_fph
    wsb = L.lib().gc_dn_attention_workspace_bytes(C.byref(d))      # head size 160 with several K/V sets: one workgroup per (query block, set)
    if wsb and not BATCH_INVARIANT:      # the library takes the set-split form only while the grid is small (nwg < 512: a function of B) and its fp32
                                         # combine rounds differently from the in-register one: batch-invariant mode never offers the workspace
        ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
        d.workspace = ws.data_ptr(); d.workspace_bytes = wsb
    L.check(L.lib().gc_dn_attention(C.byref(d), _stream()), "gc_dn_attention")
    return o


class TailDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("channels", C.c_int), ("heads", C.c_int), ("M", C.c_int64), ("rows_per_frame", C.c_int64),
                ("frames_per_half", C.c_int), ("text_len", C.c_int), ("ln_eps", C.c_float), ("attn_out", C.c_void_p), ("resid", C.c_void_p),
                ("x_in", C.c_void_p), ("out", C.c_void_p), ("w_a", C.c_void_p), ("w_kv", C.c_void_p), ("w_b", C.c_void_p),
                ("params", C.c_void_p), ("stop_after", C.c_int), ("resid_fragment_layout", C.c_int), ("in_rows", C.c_int64)]


def transformer_tail(attn_out, resid, x_in, seg_a, seg_kv, seg_b, params, heads, frames_per_half, text_len, eps=1e-5, stop_after=0, resid_frags=False,
                     halves=1):
    """Everything of a level-0 transformer block after the self-attention, one launch (gc_dn_transformer_tail): attn_out / resid / x_in
    [B, HW, 320]; seg_* / params from weights.tail_streams / weights.tail_text_stream."""
    _gpu(attn_out, resid, x_in, seg_a, seg_kv, seg_b, params)
    B, HW, Cc = attn_out.shape
    assert attn_out.is_contiguous() and resid.is_contiguous() and x_in.is_contiguous()
    if OPTIONS.ablate:
        z = _ablated("tail", attn_out.shape, attn_out.dtype, attn_out.device)
        if z is not None:
            return z
    # halves = 2 (CFG-shared prefix): the inputs hold ONE CFG half; the output has both, each continuing from the shared rows (no duplicate copies)
    out = torch.empty((halves * B, HW, Cc), dtype=attn_out.dtype, device=attn_out.device)
    d = TailDesc()
    d.dtype = _dt(attn_out); d.channels = Cc; d.heads = heads; d.M = halves * B * HW; d.rows_per_frame = HW
    d.in_rows = B * HW if halves > 1 else 0
    d.frames_per_half = frames_per_half; d.text_len = text_len; d.ln_eps = eps
    d.attn_out = attn_out.data_ptr(); d.resid = resid.data_ptr(); d.x_in = x_in.data_ptr(); d.out = out.data_ptr()
    d.w_a = seg_a.data_ptr(); d.w_kv = seg_kv.data_ptr(); d.w_b = seg_b.data_ptr(); d.params = params.data_ptr(); d.stop_after = stop_after
    d.resid_fragment_layout = int(resid_frags)
    L.check(L.lib().gc_dn_transformer_tail(C.byref(d), _stream()), "gc_dn_transformer_tail")
    return out


class HeadDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("channels", C.c_int), ("M", C.c_int64), ("rows_per_frame", C.c_int64), ("ln_eps", C.c_float),
                ("x", C.c_void_p), ("gn_coef", C.c_void_p), ("h", C.c_void_p), ("qk", C.c_void_p), ("vt", C.c_void_p), ("ldvt", C.c_int64),
                ("vt_batch_stride", C.c_int64), ("w", C.c_void_p), ("params", C.c_void_p), ("h_fragment_layout", C.c_int)]


def groupnorm_coef(x, gamma, beta, groups, eps, parts=None):
    """x [B,HW,C] -> coef fp32 [B,C,2]: GroupNorm(x)[b,:,c] = x * coef[b,c,0] + coef[b,c,1] (the statistics passes of groupnorm only;
    one tiny launch from the producer's ChanParts when given)"""
    _gpu(x)
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    if parts is not None and "gn" not in OPTIONS.ablate:
        coef = torch.empty(B, Cc, 2, dtype=torch.float32, device=x.device)
        L.check(L.lib().gc_dn_groupnorm_coef_parts(C.c_int64(B), C.c_int64(HW), Cc, groups, _p(gamma), _p(beta), C.c_float(eps), _p(parts.buf),
                                                   C.c_int64(parts.rows), parts.nslab, parts.mode, parts.col_tile, _p(coef), _stream()), "gc_dn_groupnorm_coef_parts")
        return coef
    key = (x.device, B, HW, Cc, stream_handle())
    ws = _gn_ws.get(key)
    if ws is None:
        nbytes = L.lib().gc_dn_groupnorm_workspace_bytes(C.c_int64(B), C.c_int64(HW), Cc)
        ws = _gn_ws[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    if OPTIONS.ablate:
        z = _ablated("gn", (B, Cc, 2), torch.float32, x.device)
        if z is not None:
            return z
    coef = torch.empty(B, Cc, 2, dtype=torch.float32, device=x.device)
    L.check(L.lib().gc_dn_groupnorm_coef(_dt(x), _p(x), C.c_int64(B), C.c_int64(HW), Cc, groups, _p(gamma), _p(beta), C.c_float(eps),
                                         _p(ws), _p(coef), _stream()), "gc_dn_groupnorm_coef")
    return coef


def transformer_head(x, coef, w_stream, params, eps=1e-5, h_frags=False):
    """GroupNorm apply + proj_in + LayerNorm1 + Q | K | V of a level-0 transformer block, one launch (gc_dn_transformer_head).
    x [B, HW, 320], coef from groupnorm_coef -> (h [B,HW,320], qk [B,HW,640], vt [B,320,HW])"""
    _gpu(x, coef, w_stream, params)
    B, HW, Cc = x.shape
    assert x.is_contiguous()
    if OPTIONS.ablate and "head" in OPTIONS.ablate:
        return (_ablated("head", x.shape, x.dtype, x.device), _ablated("head", (B, HW, 2 * Cc), x.dtype, x.device),
                _ablated("head", (B, Cc, HW), x.dtype, x.device))
    h = torch.empty_like(x)
    qk = torch.empty(B, HW, 2 * Cc, dtype=x.dtype, device=x.device)
    vt = torch.empty(B, Cc, HW, dtype=x.dtype, device=x.device)
    d = HeadDesc()
    d.dtype = _dt(x); d.channels = Cc; d.M = B * HW; d.rows_per_frame = HW; d.ln_eps = eps
    d.x = x.data_ptr(); d.gn_coef = coef.data_ptr(); d.h = h.data_ptr(); d.qk = qk.data_ptr(); d.vt = vt.data_ptr()
    d.ldvt = HW; d.vt_batch_stride = Cc * HW; d.w = w_stream.data_ptr(); d.params = params.data_ptr(); d.h_fragment_layout = int(h_frags)
    L.check(L.lib().gc_dn_transformer_head(C.byref(d), _stream()), "gc_dn_transformer_head")
    return h, qk, vt


def cfg_ddim_step(eps, latents, xin, guidance, cfg, alpha_t, alpha_prev, nrep):
    """eps fp32 [(2)f, H, W, ld]; latents fp32 [f,H,W,4] (in place); xin dtype [nrep*f,H,W,8] (rewritten)."""
    _gpu(eps, latents, xin)
    f = latents.shape[0]
    HW = latents.shape[1] * latents.shape[2]
    L.check(L.lib().gc_dn_cfg_ddim_step(_dt(xin), _p(eps), eps.shape[-1], C.c_int64(f), C.c_int64(HW), C.c_float(guidance),
                                        int(cfg), C.c_float(alpha_t), C.c_float(alpha_prev), _p(latents), _p(xin), nrep,
                                        _stream()), "gc_dn_cfg_ddim_step")


_disp_ws = {}


def depth_to_disparity(depth, dtype):
    """depth fp32 [H,W] -> disparity [H,W,8] (3 used) in the activation dtype: 1/(d+1e-5) / max (gc_pipeline.py:258-266)."""
    _gpu(depth)
    key = (depth.device, stream_handle())          # scratch per stream: independent chunks may run concurrently
    ws = _disp_ws.get(key)
    if ws is None:
        ws = _disp_ws[key] = torch.zeros(1, dtype=torch.int32, device=depth.device)
    d = depth.contiguous()
    out = torch.empty(d.shape[0], d.shape[1], 8, dtype=dtype, device=d.device)
    L.check(L.lib().gc_dn_depth_to_disparity(DT[dtype], _p(d), C.c_int64(d.numel()), _p(out), _p(ws), _stream()),
            "gc_dn_depth_to_disparity")
    return out


def mask_composite(edited_hwc, unedited_hwc=None, mask_hw=None):
    """edited fp32 [H,W,C>=3] (channels-last), unedited fp32 [H,W,3], mask fp32 [H,W] -> fp32 [H,W,3] (gc_pipeline.py:226-234)."""
    _gpu(edited_hwc, unedited_hwc, mask_hw)
    H, W = edited_hwc.shape[0], edited_hwc.shape[1]
    out = torch.empty(H, W, 3, dtype=torch.float32, device=edited_hwc.device)
    L.check(L.lib().gc_dn_mask_composite(_p(edited_hwc), edited_hwc.shape[-1], _p(unedited_hwc), _p(mask_hw), _p(out),
                                         C.c_int64(H * W), _stream()), "gc_dn_mask_composite")
    return out
