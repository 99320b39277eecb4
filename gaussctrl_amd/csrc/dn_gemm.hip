// dn_gemm.hip -- host side of the bf16 / f16 / fp8 MFMA GEMM and implicit-GEMM 3x3 convolution of the SD1.5 UNet / ControlNet / VAE
// blocks (gfx950): problem validation, kernel selection, launches.  The kernels live in dn_gemm_kernels.h and are instantiated in
// three translation units that compile in parallel: dn_gemm_plain.hip (lean epilogue), dn_gemm_fuse.hip (fused-normalisation
// epilogue), dn_gemm_fp8.hip (e4m3 operands on the block-scaled MFMA).  Reference call sites: see dn_gemm_kernels.h.
#include "dn_gemm_kernels.h"

namespace {

int choose_splits(int64_t blocks, int nk)
{
    // split only long-K / few-tile problems (3x3 convs on 16x16 and 8x8 maps): each slice keeps >= 8 k-tiles and the
    // fp32 partial slabs ([S][M][N], written once, read once) stay small next to the weight stream
    if (blocks >= 128 || nk < 32) return 1;
    int s = (int)((448 + blocks - 1) / blocks);
    if (s > nk / 8) s = nk / 8;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

// 8-wave kernel: pick the m-tiles per wave (workgroup rows = 64 MT); 0 = use the 4-wave kernel.  One workgroup per CU, so the
// grid runs in ceil(tiles / 256) rounds and a tile costs about (MT + 1) units (MT of MFMA work + fill / epilogue): minimise
// rounds * (MT + 1).  Matches the per-shape sweep in profiles/ (scripts/bench_kernels.py with GC_GEMM_MT=2|3|4).
int choose_mt(int64_t M, int64_t N, int ntw, bool forced)
{
    const int64_t nbn = (N + 32 * ntw - 1) / (32 * ntw);
    if (((M + 127) / 128) * nbn < 96 && !forced) return 0;      // small problems: 4-wave kernel (+ split-K)
    int best = 2;
    int64_t best_cost = -1;
    for (int mt = 2; mt <= 4; ++mt) {
        const int64_t tiles = ((M + 64 * mt - 1) / (64 * mt)) * nbn;
        const int64_t cost = ((tiles + 255) / 256) * (mt + 1);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && mt == 4)) { best = mt; best_cost = cost; }
    }
    return best;
}

// Column-panel width of the tile order (tile_coords in dn_gemm_kernels.h).  The `conc` workgroups an XCD runs together execute a contiguous run of
// logical tiles = a patch of rows x cols tiles; over the fabric the XCD fetches rows A panels (bm rows x the k range; a 3 x 3 conv's im2col panel is
// 9 x its real bytes, the rest are L2 hits) and cols W panels (bn rows x the k range).  Pick the width with the fewest bytes; ties keep whole rows.
int choose_pw(int64_t nbm, int64_t nbn, int64_t total_wgs, int conc, int64_t bm, int64_t bn, bool conv)
{
    int64_t r = total_wgs / 8 < 1 ? 1 : total_wgs / 8;
    if (r > conc) r = conc;
    const int64_t tiles = nbm * nbn;
    if (r >= tiles) return 0;                                   // an XCD covers a whole slice anyway
    const int64_t ca = conv ? bm : 9 * bm, cw = 9 * bn;         // panel bytes up to a common factor (x 9: integers)
    int best = 0;
    int64_t best_cost = -1;
    for (int64_t pw = nbn; pw >= 1; --pw) {
        const int64_t per = nbm * pw;
        const int64_t cols = std::min<int64_t>(nbn, pw * ((r + per - 1) / per)), rows = std::min<int64_t>(nbm, (r + pw - 1) / pw);
        const int64_t cost = rows * ca + cols * cw;
        if (best_cost < 0 || cost < best_cost) { best = pw == nbn ? 0 : (int)pw; best_cost = cost; }
    }
    return best;
}

// kernel_variant bits 16..23: n + 1 forces panel width n (0 = whole rows) -- experiments / the bit-identity test
int pw_of(const gc_gemm_desc *d, int64_t nbm, int64_t nbn, int64_t total_wgs, int conc, int64_t bm, int64_t bn)
{
    const int f = (d->kernel_variant >> 16) & 0xff;
    if (f) return f - 1;
    return choose_pw(nbm, nbn, total_wgs, conc, bm, bn, d->mode == 1);
}

void plan(const gc_gemm_desc *d, int *ntw, int *splits, int *tps)
{
    // GEGLU pairs tiles (nt, nt+1) inside a wave: needs an even number of n-tiles per wave
    *ntw = (d->N % 160 == 0 && d->N % 128 != 0 && !d->geglu) ? 5 : 4;
    if (d->softmax_keys > 0) *ntw = 5;                                    // softmax-heads epilogue: one 80-column head block per wave column
    const int bn = 32 * *ntw;
    const int64_t Msel = d->plan_rows > 0 ? d->plan_rows : d->M;         // batch-invariant planning: the rows of one frame
    const int64_t blocks = ((Msel + BM - 1) / BM) * ((d->N + bn - 1) / bn);
    const int nk = (int)((d->K + BK - 1) / BK);
    *splits = (d->geglu || d->w_set_rows > 0 || d->softmax_keys > 0) ? 1 : choose_splits(blocks, nk);
    *tps = (nk + *splits - 1) / *splits;
    *splits = (nk + *tps - 1) / *tps;
}


// fp8 (k_gemm8q) problems: column tile, m-tiles per wave and k-slices.  Long-K plain problems on part-filled grids (3x3 convs on 16 x 16
// maps, the FF down projection on few rows) are cut into k-slices of >= 8 k-tiles (1 024 elements) so that ~256 workgroups run; the
// split-K reduce kernels of the 2-byte path finish them (and leave the GroupNorm partials).  assume_ws: size query before the caller
// has a workspace.
struct SelQ { int ntw, mt, splits, tps; };
void select_fp8(const gc_gemm_desc *d, SelQ *o, bool assume_ws)
{
    const int force_mt = d->kernel_variant & 7;
    o->ntw = (d->N % 160 == 0 && d->N % 128 != 0 && !d->geglu) ? 5 : 4;      // (GEGLU pairs n-tiles inside a wave: even count)
    // the e4m3 fragments are 8 registers each (32 k per lane): only the variants that stay under 256 VGPRs without spilling are
    // instantiated -- (NTW 5, MT 2), (NTW 4, MT 2 | 3); a spill reload is a VM load that stalls behind the LDS-DMA queue
    int mt = force_mt ? force_mt : choose_mt(d->M, d->N, o->ntw, true);
    if (o->ntw == 5) mt = 2; else if (mt > 3) mt = 3;
    if (mt < 2) mt = 2;
    const int nkq = (int)(d->K / 128);
    o->splits = 1; o->tps = nkq;
    const bool plain = !d->geglu && !d->out_t && !d->out_fp8 && !d->out_f32 && d->out && !d->ln_row_stats && !d->out_row_stats && !d->out_group_stats && d->act == 0;
    // (plan_rows: the k-slices -- the accumulation order of every output row -- follow the rows ONE frame contributes, as in the 2-byte path)
    const int64_t Msel = d->plan_rows > 0 ? d->plan_rows : d->M;
    const int64_t nbn = (d->N + 32 * o->ntw - 1) / (32 * o->ntw), tiles = ((Msel + 127) / 128) * nbn;
    if (plain && !force_mt && !(d->kernel_variant & 0x40) && tiles <= 128 && nkq >= 16) {
        int s = (int)std::min<int64_t>(tiles < 96 ? (256 + tiles - 1) / tiles : 256 / tiles, nkq / 8);
        if (s > 16) s = 16;
        if (s >= 2 && (assume_ws || (d->workspace && d->workspace_bytes >= sizeof(float) * (size_t)s * (size_t)d->M * (size_t)d->N))) {
            mt = 2; o->tps = (nkq + s - 1) / s; o->splits = (nkq + o->tps - 1) / o->tps;
        }
    }
    o->mt = mt;
}
}  // namespace

extern "C" size_t gc_dn_gemm_workspace_bytes(const gc_gemm_desc *d)
{
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    if (d->fp8) {
        if (d->K % 128 != 0) return 0;
        SelQ q;
        select_fp8(d, &q, true);
        return q.splits > 1 ? sizeof(float) * (size_t)q.splits * (size_t)d->M * (size_t)d->N : 0;
    }
    int ntw, splits, tps;
    plan(d, &ntw, &splits, &tps);
    return splits > 1 ? sizeof(float) * (size_t)splits * (size_t)d->M * (size_t)d->N : 0;
}

namespace {
// lean LayerNorm fold (dn_gemm_ln.hip): 1 = this problem is a producer of row partials on the lean epilogue, 2 = a LayerNorm-folded consumer,
// 0 = neither (or kernel_variant bit 0x800 asks for the round-2 FUSE epilogue: A/B and tests)
int ln_lean_kind(const gc_gemm_desc *d)
{
    if ((d->kernel_variant & 0x800) || d->fp8 || d->mode != 0 || d->K % 64 != 0 || d->out_group_stats || d->out_chan_parts || d->N < 4) return 0;
    if (d->out_row_stats && !d->ln_row_stats)
        return (!d->geglu && d->act == 0 && !d->out_f32 && !d->out_t && d->out && (!d->rowvec || d->rows_per_batch >= 256)) ? 1 : 0;
    if (d->ln_row_stats && !d->out_row_stats) return d->softmax_keys > 0 ? 3 : 2;
    return 0;
}
// kernel choice of one problem (shared by the launcher and the row-statistics layout query)
struct Sel { int mode, ntw, splits, tps, mt8; };     // mt8 > 0: 8-wave kernel with MT = mt8; 0: 4-wave kernel
int select(const gc_gemm_desc *d, Sel *o, bool want_parts)
{
    int mode = 0;
    if (d->mode == 1) mode = (d->Cin % 64 == 0) ? 2 : 1;
    int ntw, splits, tps;
    plan(d, &ntw, &splits, &tps);
    if (splits > 1 && (!d->workspace || d->workspace_bytes < sizeof(float) * (size_t)splits * (size_t)d->M * (size_t)d->N)) {
        splits = 1; tps = (int)((d->K + BK - 1) / BK);      // no workspace: run unsplit
    }
    const int bn = 32 * ntw;
    const int64_t nbn = (d->N + bn - 1) / bn;
    const int nk_host = (int)((d->K + BK - 1) / BK);
    // kernel selection overrides travel in the descriptor (tests / experiments); 0 = automatic.  No process-wide state.
    const int kv = d->kernel_variant;
    const int force_mt = kv & 7;                                   // 2 | 3 | 4: force the 8-wave kernel's m-tiles per wave
    const int use8 = (kv & 0x10) ? 0 : ((kv & 0x20) ? 2 : 1);      // 0x10: 4-wave kernel only; 0x20: force the 8-wave kernel
    const int convsplit = (kv & 0x40) ? 0 : ((kv & 0x80) ? 1 : 2); // 0x40: no k-slices for part-filled conv grids; 0x80: not for the small ones (8x8 maps, stride-2 convs: 4-wave split-K kernel, the round-2..4 choice)
    o->mode = mode; o->ntw = ntw; o->splits = splits; o->tps = tps; o->mt8 = 0;
    // 8-wave LDS-DMA kernel (one workgroup per CU, software-pipelined): workgroup tile (64 MT) x (32 NTW).
    if (use8 && d->zeros) {
        // (plan_rows: the kernel family and the k-slices follow the per-frame problem; the m-tiles per wave, which do not change any
        // accumulation order, follow the real grid)
        const int64_t Msel = d->plan_rows > 0 ? d->plan_rows : d->M;
        int mt = force_mt ? force_mt : choose_mt(Msel, d->N, ntw, use8 == 2);
        if (mt && !force_mt && Msel != d->M) mt = choose_mt(d->M, d->N, ntw, true);
        const int64_t tiles8 = ((Msel + 127) / 128) * nbn;
        // long-K problems with a part-filled grid (16x16-map convs, the 5120 -> 1280 FF projection): k-slices of >= 12 k-tiles
        // fill the CUs, the small grids too (M = 384: 30 tiles x 8 slices, every XCD owns one slice -- round 5: 34.4 -> 25.3 us per 8 x 8-map conv,
        // profiles/r05_conv8x8_slices_ab.txt); variant 0x40 sends such convs back to the 4-wave split-K kernel, 0x80 only the small ones
        const bool small = tiles8 < 96;
        if (!mt && mode == 0 && !force_mt) mt = 2;       // small linears: the 8-wave kernel's fill + epilogue is the shorter one (12.6 vs 20.9 us at M = 384)
        // part-filled single-round grids of short-K linears: 64-row tiles, two workgroups per CU (all resident when <= 512 tiles)
        if (mode == 0 && !force_mt && ntw == 4 && d->K % 64 == 0 && !d->geglu && !d->out_t && !(kv & 0x100) &&
            (!(d->ln_row_stats || d->out_row_stats || d->out_group_stats) || ln_lean_kind(d)) && !want_parts) {
            const int64_t t1 = ((d->M + 63) / 64) * nbn, t2 = ((d->M + 127) / 128) * nbn;
            if (mt == 2 && t2 <= 256 && t1 <= 512 && t1 > 128 && nk_host <= 24) mt = 1;      // (same accumulation order as MT 2)
        }
        if (d->w_set_rows > 0 && mt) {       // weight sets: the row tile must divide the rows of a set (a tile never straddles two sets)
            int pick = 0;
            const int cand[5] = {mt, 4, 3, 2, (ntw == 4 && d->softmax_keys == 0) ? 1 : 2};
            for (int c = 0; c < 5 && !pick; ++c) if (d->w_set_rows % (64 * cand[c]) == 0) pick = cand[c];
            mt = pick ? pick : mt;
        }
        const bool want_split = !force_mt && use8 != 2 && !d->geglu && d->w_set_rows == 0 && d->softmax_keys == 0 && d->workspace && tiles8 <= 128 && (mode == 0 ? (!small || nk_host >= 24) : (convsplit >= (small ? 2 : 1)));
        int s8 = 1, tps8 = nk_host;
        if (want_split) {
            const int smt = (kv >> 24) & 7;                                  // experiments: forced m-tiles per wave of the k-sliced problems
            // 16 x 16-map convs: 192-row tiles x 3 slices (60.4 vs 63.7 us); the small grids and the linears: 128 rows; round 6: the 8 x 8-map convs of a 4-chunk launch
            // set (M = 1 536 = 24 whole images) 256-row tiles x 4 slices = four whole images per tile (profiles/r06_gemm_mt_scan_launch_sets.txt: 73.4 -> 65.8 us, 126.7 -> 111.4 us)
            const bool img8 = mode != 0 && !small && d->Ho * d->Wo <= 64 && Msel % 256 == 0;
            const int mts = (smt >= 2 && smt <= 4) ? smt : (img8 ? 4 : ((mode != 0 && !small) ? 3 : 2));
            const int64_t tiles_s = ((Msel + 64 * mts - 1) / (64 * mts)) * nbn;
            s8 = (int)std::min<int64_t>(std::max<int64_t>(256 / tiles_s, 2), nk_host / 12);         // one round: tiles x slices <= 256 workgroups
            if (s8 >= 2 && d->workspace_bytes >= sizeof(float) * (size_t)s8 * (size_t)d->M * (size_t)d->N) {
                mt = mts; tps8 = (nk_host + s8 - 1) / s8; s8 = (nk_host + tps8 - 1) / tps8;
            } else s8 = 1;
        }
        if (splits > 1 && s8 == 1 && !force_mt && use8 != 2) mt = 0;      // otherwise: 4-wave split-K kernel as planned
        if (mt) { o->mt8 = mt; o->splits = s8; o->tps = tps8; }
    }
    return 0;
}
}  // namespace

// number of column slabs gc_dn_gemm writes per row into desc->out_row_stats (layout [slots][M][2]) for this problem
extern "C" int gc_dn_gemm_row_stat_slots(const gc_gemm_desc *d)
{
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    if (d->fp8) { const int ntw = (d->N % 160 == 0 && d->N % 128 != 0) ? 5 : 4; return (int)((d->N + 16 * ntw - 1) / (16 * ntw)); }
    Sel sel;
    select(d, &sel, false);
    return sel.splits > 1 ? (int)((d->N / 4 + 63) / 64) : (int)((d->N + 16 * sel.ntw - 1) / (16 * sel.ntw));
}

namespace {
// channel-partial layout of one problem: rows per slab (0 = unsupported) and slab slots per batch
void chan_parts_layout(const gc_gemm_desc *d, const Sel &sel, int64_t *rows, int *nslab, int *col_tile)
{
    *rows = 0; *nslab = 0; *col_tile = 0;
    const int64_t rpb = d->rows_per_batch;
    if (d->fp8) {                    // k_gemm8q: only the k-sliced problems (their reduce kernel is the 2-byte path's)
        SelQ q;
        select_fp8(d, &q, false);
        if (d->K % 128 != 0 || rpb < 256 || rpb % 32 != 0 || d->M % rpb != 0 || d->gn_groups < 1 || d->N % d->gn_groups != 0) return;
        const int64_t cpg = d->N / d->gn_groups;
        if (q.splits > 1) {          // the reduce-epilogue kernel produces them (64-column blocks)
            if (cpg > 64) return;
            *rows = CS_RB; *nslab = (int)(rpb / CS_RB); *col_tile = 64;
            return;
        }
        // unsliced fast convs: k_gemm8q's own channel-partial epilogue (lean: bias / row vector / scale / residual / 2-byte store)
        const bool lean = d->mode == 1 && !d->upsample && !d->geglu && !d->out_t && !d->out_fp8 && !d->out_f32 && d->out && d->act == 0 &&
                          !d->ln_row_stats && !d->out_row_stats && !d->out_group_stats;
        const int64_t bm = 64 * q.mt;
        if (!lean || bm > rpb || cpg > 32 * q.ntw) return;
        *rows = bm;
        *nslab = (int)(rpb % bm == 0 ? rpb / bm : (rpb + bm - 1) / bm + 1);
        *col_tile = 32 * q.ntw;
        return;
    }
    if (d->geglu || d->act != 0 || d->out_t || !d->out || d->out_f32 || d->ln_row_stats || d->out_row_stats || d->out_group_stats) return;
    if (rpb < 256 || rpb % 32 != 0 || d->M % rpb != 0) return;
    if (d->gn_groups < 1 || d->N % d->gn_groups != 0) return;
    const int64_t cpg = d->N / d->gn_groups;
    if (sel.splits > 1) {            // the reduce-epilogue kernel produces them (64-column blocks)
        if (cpg > 64) return;
        *rows = CS_RB; *nslab = (int)(rpb / CS_RB); *col_tile = 64;
        return;
    }
    if (!sel.mt8) return;
    if ((d->mode == 1 && d->upsample) || (d->mode == 0 && d->K % 64 != 0)) return;
    const int64_t bm = 64 * (sel.mt8 < 2 ? 2 : sel.mt8);
    if (bm > rpb || cpg > 32 * sel.ntw) return;
    *rows = bm;
    *nslab = (int)(rpb % bm == 0 ? rpb / bm : (rpb + bm - 1) / bm + 1);
    *col_tile = 32 * sel.ntw;
}
}  // namespace

extern "C" int gc_dn_gemm_chan_parts_layout(const gc_gemm_desc *d, int64_t *rows_per_slab, int *nslab, int *col_tile)
{
    GC_REQUIRE(d && rows_per_slab && nslab && col_tile, "null argument");
    *rows_per_slab = 0; *nslab = 0; *col_tile = 0;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->rows_per_batch <= 0) return GC_OK;
    Sel sel;
    select(d, &sel, true);
    chan_parts_layout(d, sel, rows_per_slab, nslab, col_tile);
    return GC_OK;
}

extern "C" int gc_dn_gemm(const gc_gemm_desc *d, void *stream)
{
    GC_REQUIRE(d && d->W && d->A, "null descriptor / operand");
    GC_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "empty problem");
    GC_REQUIRE(d->K % 8 == 0, "K must be a multiple of 8 (pad channels on the host)");
    GC_REQUIRE(d->N % 4 == 0, "N must be a multiple of 4 (pad output channels on the host)");
    GC_REQUIRE(d->dtype == DT_BF16 || d->dtype == DT_F16, "dtype must be 0 (bf16) or 1 (f16)");
    GemmArgs g;
    g.pw = 0;
    g.M = d->M; g.N = d->N; g.K = d->K; g.A = d->A; g.lda = d->lda;
    g.B = d->B; g.Hi = d->Hi; g.Wi = d->Wi; g.Cin = d->Cin; g.Ho = d->Ho; g.Wo = d->Wo; g.stride = d->stride; g.ups = d->upsample; g.pad = d->pad_lo;
    g.W = d->W; g.bias = d->bias; g.rowvec = d->rowvec; g.ld_rowvec = d->ld_rowvec;
    g.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : 1;
    g.residual = d->residual; g.ldr = d->ldr; g.out_scale = d->out_scale; g.act = d->act; g.geglu = d->geglu;
    g.out = d->out; g.ldc = d->ldc; g.out_f32 = d->out_f32; g.out_t = d->out_t; g.ldt = d->ldt; g.t_batch_stride = d->t_batch_stride; g.t_col0 = d->t_col0;
    g.row_stats = d->ln_row_stats; g.row_stat_slots = d->ln_row_stat_slots > 0 ? d->ln_row_stat_slots : 1;
    g.colsum = d->ln_colsum; g.ln_eps = d->ln_eps; g.ln_inv_k = 1.f / (float)d->K;
    g.out_row_stats = d->out_row_stats; g.out_group_stats = d->out_group_stats; g.gn_groups = d->gn_groups;
    g.gn_cpg = d->gn_groups > 0 ? (int)(d->N / d->gn_groups) : 1;
    g.w_scale = (const unsigned char *)d->w_scale; g.a_scale = d->a_scale;
    g.out_fp8 = d->out_fp8; g.out_qscale = d->out_fp8 > 0 ? exp2f((float)(127 - d->out_fp8)) : 1.f;
    GC_REQUIRE(d->out_fp8 >= 0 && d->out_fp8 < 255, "out_fp8 must be 0 (off) or an E8M0 byte 1 .. 254");
    GC_REQUIRE(!d->out_fp8 || (d->fp8 && d->mode == 0 && d->out && !d->out_f32 && !d->out_t && !d->out_chan_parts && !d->ln_row_stats && !d->out_row_stats &&
                               !d->out_group_stats && d->ldc % 4 == 0),
               "out_fp8: an fp8 linear with a plain output (no fp32 / transposed output, no statistics, no LayerNorm fold), ldc % 4 == 0");
    g.dbg = (d->kernel_variant >> 8) & 0x0f;
    g.conv_korder = (d->kernel_variant & 0x1000) ? 0 : 1;
    g.w_set_rows = d->w_set_rows; g.w_set_stride = d->w_set_stride; g.sm_keys = d->softmax_keys;
    GC_REQUIRE(d->w_set_rows >= 0 && d->softmax_keys >= 0 && d->softmax_keys <= 80, "bad weight-set / softmax arguments");
    GC_REQUIRE(!d->softmax_keys || (d->ln_row_stats && d->N % 80 == 0 && d->out && !d->out_f32 && !d->out_t && !d->geglu && !d->residual && d->act == 0),
               "softmax_keys: a LayerNorm-folded linear with N % 80 == 0 and a plain 2-byte output");
    GC_REQUIRE((d->ln_row_stats == nullptr) == (d->ln_colsum == nullptr), "ln_row_stats and ln_colsum must be given together");
    GC_REQUIRE(!d->ln_row_stats || d->mode == 0, "LayerNorm folding applies to linear GEMMs");
    GC_REQUIRE(!(d->out_row_stats || d->out_group_stats) || (!d->geglu && !d->out_t && d->out), "output statistics need a plain (non-GEGLU, non-transposed) output");
    GC_REQUIRE(!d->out_group_stats || (g.rows_per_batch % 16 == 0 && d->gn_groups >= 1 && d->gn_groups <= GS_MAXG && d->N % d->gn_groups == 0),
               "out_group_stats needs rows_per_batch % 16 == 0 and 1 <= gn_groups <= 32 dividing N");
    if (d->mode == 1) {
        GC_REQUIRE(d->Cin % 8 == 0 && d->K == 9 * (int64_t)d->Cin, "conv3x3: K must be 9*Cin with Cin % 8 == 0");
        GC_REQUIRE(d->M == (int64_t)d->B * d->Ho * d->Wo, "conv3x3: M must be B*Ho*Wo");
        GC_REQUIRE(d->pad_lo == 0 || d->pad_lo == 1, "conv3x3: pad_lo must be 0 or 1");
    } else {
        GC_REQUIRE(d->lda >= d->K && d->lda % 8 == 0, "linear: lda must be >= K and a multiple of 8");
    }
    // the kernels index the operands with 32-bit element offsets (per-lane DMA offsets are one register): refuse what does not fit instead of faulting
    // (a 42-view VAE decode batch at 512 x 512 x 256 channels is 2.8 G elements: split the batch on the host)
    GC_REQUIRE((d->mode == 1 ? (int64_t)d->B * d->Hi * d->Wi * d->Cin : d->M * d->lda) < ((int64_t)1 << 31) && (int64_t)d->N * d->K < ((int64_t)1 << 31),
               "operand too large for 32-bit element offsets: split the batch");
    if (d->geglu) GC_REQUIRE(d->N % 32 == 0 && !d->out_t, "geglu needs N % 32 == 0");
    const int force_mt = d->kernel_variant & 7;
    GC_REQUIRE(force_mt >= 0 && force_mt <= 4, "kernel_variant: MT must be 0 .. 4");
    Sel sel;
    select(d, &sel, d->out_chan_parts != nullptr);
    g.chan_parts = d->out_chan_parts; g.cp_nslab = 0; g.cp_rows = 0;
    if (d->out_chan_parts) {
        int64_t rows; int ns, ct;
        chan_parts_layout(d, sel, &rows, &ns, &ct);
        GC_REQUIRE(rows > 0, "out_chan_parts: this problem cannot produce channel partials (see gc_dn_gemm_chan_parts_layout)");
        g.cp_nslab = ns; g.cp_rows = (int)rows;
        if (sel.mt8 == 1) sel.mt8 = 2;
    }
    if (fuse_of(g) && sel.mt8) {     // the 8-wave kernel carries the fused epilogue for conv (generic / fast) and K % 64 == 0 linears only
        const bool upsampled = d->mode == 1 && d->upsample, ragged = d->mode == 0 && d->K % 64 != 0;
        GC_REQUIRE(!upsampled && !ragged, "row / group statistics and LayerNorm folding: not with upsample-fused convs or K % 64 != 0 linears");
    }
    hipStream_t s = gc::S(stream);
    g.zeros = d->zeros;
    if (d->fp8) {      // OCP e4m3 operands on the block-scaled MFMA (k_gemm8q): A / W are bytes, K counts fp8 elements
        GC_REQUIRE(d->zeros && d->w_scale, "fp8: zeros page and per-row weight scales are required");
        GC_REQUIRE(d->K % 128 == 0, "fp8: K % 128 == 0");
        GC_REQUIRE(!(d->geglu || d->out_t) || (d->mode == 0 && !fuse_of(g)), "fp8: GEGLU / transposed output on linears with the plain epilogue only");
        GC_REQUIRE(d->mode == 0 || (d->Cin % 128 == 0 && !d->upsample), "fp8 conv: Cin (padded) % 128 == 0, no fused upsample");
        GC_REQUIRE(d->mode == 1 || d->lda % 16 == 0, "fp8 linear: lda % 16 == 0");
        GC_REQUIRE((int64_t)d->N * d->K < ((int64_t)1 << 31) && (d->mode == 1 ? (int64_t)d->B * d->Hi * d->Wi * d->Cin : d->M * d->lda) < ((int64_t)1 << 31), "fp8: 32-bit offsets");
        SelQ q;
        select_fp8(d, &q, false);
        const int ntw = q.ntw, mt = q.mt;
        g.splits = q.splits; g.tiles_per_split = q.tps; g.ws = q.splits > 1 ? (float *)d->workspace : nullptr;
        GC_REQUIRE(!d->out_chan_parts || q.splits > 1 || (d->mode == 1 && !fuse_of(g)), "fp8: channel partials come from k-sliced problems and fast convs");
        const int64_t nbn_q = (d->N + 32 * ntw - 1) / (32 * ntw), nbm_q = (d->M + 64 * mt - 1) / (64 * mt);
        const dim3 gq((unsigned)(nbm_q * nbn_q), (unsigned)q.splits);
        g.pw = pw_of(d, nbm_q, nbn_q, nbm_q * nbn_q * q.splits, 32, 64 * mt, 32 * ntw);
        dn_gemm_launch_fp8(g, d->dtype, d->mode == 1 ? 2 : 3, ntw, mt, gq, s);
        if (q.splits > 1) { if (g.chan_parts) dn_gemm_launch_splitk_epilogue_cs(g, d->dtype, s); else dn_gemm_launch_splitk_epilogue(g, d->dtype, s); }
        return gc::check_launch("gc_dn_gemm(fp8)");
    }
    g.splits = sel.splits; g.tiles_per_split = sel.tps; g.ws = (float *)d->workspace;
    const int bn = 32 * sel.ntw;
    const int64_t nbn = (d->N + bn - 1) / bn;
    g.persist = 0;
    const int lnk = (sel.mt8 && sel.splits == 1) ? ln_lean_kind(d) : 0;
    if (d->w_set_rows > 0 || d->softmax_keys > 0) {
        GC_REQUIRE(lnk != 0, "weight sets / softmax_keys need a lean LayerNorm-fold problem (K % 64 == 0 linear with ln_row_stats or out_row_stats, no forced variant)");
        GC_REQUIRE(d->w_set_rows == 0 || (d->w_set_rows % (64 * sel.mt8) == 0 && d->M % d->w_set_rows == 0), "w_set_rows must be a multiple of the row tile and divide M");
    }
    if (sel.mt8 == 4 && sel.ntw == 4 && sel.mode == 0 && sel.splits == 1 && d->K % 64 == 0 && d->K <= 64 * 24 && (!fuse_of(g) || lnk == 2) && !g.chan_parts &&
        !g.rowvec && !g.out_t && !g.out_f32 && g.out && g.act != 2 && !(d->kernel_variant & 0x200)) {
        // multi-round short-K linear (the GEGLU FF-up projections): persistent workgroups, next tile's fill under this tile's epilogue
        const int64_t tiles = ((d->M + 255) / 256) * nbn;
        if (tiles > 256) g.persist = 256;
    }
    if (sel.mt8) {
        const int64_t nbm8 = (d->M + 64 * sel.mt8 - 1) / (64 * sel.mt8);
        const dim3 grid8((unsigned)(nbm8 * nbn), (unsigned)sel.splits);
        g.pw = pw_of(d, nbm8, nbn, g.persist ? 8 * 32 : nbm8 * nbn * sel.splits, sel.mt8 == 1 ? 64 : 32, 64 * sel.mt8, bn);
        if (lnk) dn_gemm_launch_ln(g, d->dtype, lnk, lean_of(g, sel.mode) && g.persist == 0, sel.ntw, sel.mt8, grid8, s);
        else if (fuse_of(g)) dn_gemm_launch_fuse(g, d->dtype, sel.mode, sel.ntw, sel.mt8 < 2 ? 2 : sel.mt8, grid8, s);
        else if (g.chan_parts && sel.splits == 1) dn_gemm_launch_cs(g, d->dtype, sel.mode, sel.ntw, sel.mt8, grid8, s);
        else if (lean_of(g, sel.mode) && g.persist == 0 && !(d->kernel_variant & 0x400)) dn_gemm_launch_lean(g, d->dtype, sel.mode, sel.ntw, sel.mt8, grid8, s);
        else dn_gemm_launch_plain(g, d->dtype, sel.mode, sel.ntw, sel.mt8, grid8, s);
    } else {
        const int64_t nbm = (d->M + BM - 1) / BM;
        const dim3 grid((unsigned)(nbm * nbn), (unsigned)sel.splits);
        g.pw = pw_of(d, nbm, nbn, nbm * nbn * sel.splits, 64, BM, bn);
        if (fuse_of(g)) dn_gemm_launch_fuse(g, d->dtype, sel.mode, sel.ntw, 0, grid, s); else dn_gemm_launch_plain(g, d->dtype, sel.mode, sel.ntw, 0, grid, s);
    }
    if (sel.splits > 1) { if (g.chan_parts) dn_gemm_launch_splitk_epilogue_cs(g, d->dtype, s); else dn_gemm_launch_splitk_epilogue(g, d->dtype, s); }
    return gc::check_launch("gc_dn_gemm");
}
