// raster_sort.hip -- hand-written two-level tile binning for the splat rasterizer (gfx950).
//
// Replaces the "sort M 64-bit keys (tile<<32 | depth)" step of gsplat 0.1.3's bin_and_sort_gaussians (reached from
// /root/reference/gaussctrl/gc_model.py:174-186,191-202) with an equivalent that moves ~4x fewer bytes:
//
//   1. stable LSD radix sort of the N Gaussians by depth bits (4 passes x 8 bits over (key32, id32) pairs; culled Gaussians carry
//      key 0xFFFFFFFF and sink to the end).  Ties keep ascending Gaussian id.
//   2. tiles-hit gathered in depth order -> scan -> emission offsets; every visible Gaussian emits (tile id, gaussian id) for the
//      tiles of its box IN DEPTH ORDER.
//   3. ONE or TWO stable radix passes over the M (tile id, gaussian id) pairs (8 bits each; 1024 tiles -> 2 passes).
//      Stable-by-tile of a depth-ordered sequence == sort by (tile, depth, id): exactly the order a stable sort of the 64-bit keys gives,
//      so gaussian_ids_sorted / tile_bins stay bit-identical to the oracle.
//   4. tile bins from the sorted tile ids; the 64-bit keys are re-assembled only on request (tests / API compatibility).
//
// Radix pass = histogram kernel (256 LDS counters / workgroup) -> scan of the [digit][workgroup] table (the wave64 scan of
// raster_bin.hip) -> scatter kernel with a STABLE in-workgroup rank: 4096 items / workgroup, each wave walks its contiguous 1024 in 16
// rounds of 64; the rank among equal digits = a per-wave running digit counter in LDS (advanced by the leader lane of each digit group,
// 8 ballots (match-any) + popcount inside the round) + a prefix over the four waves per digit, seeded with the global digit / workgroup
// offset (round 6: 8 KB of LDS; rounds 1-5 kept a 64 KiB (round, wave) x digit table and ran two workgroups per CU).
#include "common.h"

// raster_bin.hip (internal, not in the public header): scan of C views' tile counts, optionally gathered through `order` first;
// the three scan kernels run with grid.y = view on per-view scratch regions `ws_view_stride` bytes apart
extern "C" int gc_raster_scan_tiles_views(int64_t N, int C, const int32_t *in, const int32_t *order, int32_t *out, int32_t *count_dev,
                                          void *workspace, size_t workspace_bytes, int64_t ws_view_stride, void *stream);
extern "C" size_t gc_raster_scan_workspace_bytes(int64_t N);

namespace {

constexpr int TILE = 16;
constexpr int RT = 256;            // threads
constexpr int RI = 16;             // items per thread
constexpr int RB = RT * RI;        // 4096 items per workgroup

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// n_dev (optional): the item count lives on the device (sync-free binning); `n` is then the capacity it is clamped to
__device__ __forceinline__ int64_t live_count(int64_t n, const int32_t *n_dev)
{
    if (!n_dev) return n;
    const int64_t m = (int64_t)*n_dev;
    return m < n ? m : n;
}

// Scan of a radix pass's [digit][workgroup] table in ONE launch: every workgroup scans its 2048 entries and publishes its total; the last
// one to arrive (ticket) turns the totals into exclusive offsets in place.  Consumers add sums[entry / 2048] themselves -- the generic
// three-kernel scan (local / sums / add, raster_bin.hip) costs three launches of 5-8 us for a 1 MB table, seven times per view.
constexpr int TS_CHUNK = 2048;

// Batched views (round 5: gc_raster_depth_order_views / gc_raster_bin_tiles_views).  Every kernel of this file takes the view (camera) index
// from blockIdx.y: its workspace arrays (keys / vals / digit tables / scan sums / ticket) live in a per-view region of `ws` 4-byte words, the
// external per-view arrays have their own strides.  A single-view launch has gridDim.y == 1 and every offset is 0.
struct VS {
    int64_t ws;        // workspace region per view, in 4-byte words
    int64_t ext;       // external per-view output / input arrays ([C][n] int32: depth_order, gaussian_ids_sorted): elements per view
    int64_t src;       // per-Gaussian source arrays ([C][N]: depths, radii, num_tiles_hit, tile boxes, xys / 2)
    int nd;            // stride of the device-side counts (n_dev / overflow): 1 per view (0 when unused)
};
__device__ __forceinline__ int block_incl_scan256(int v, int *total, int *wsum /* LDS[4] */)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int sc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(sc, d, 64); if (lane >= d) sc += y; }
    __syncthreads();
    if (lane == 63) wsum[wid] = sc;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) if (w < wid) off += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return sc + off;
}

// Zero-initialisation of a launch set's counters in ONE launch (round 6): the per-view tickets / first-pass digit tables / tile_bins / overflow
// flags were one hipMemsetAsync each -- 27 fills of ~5 us per launch set of 8 views, 4 % of the raster-only chain at 1 M Gaussians
// (profiles/r06_raster_kernel_stats_1000000_before_clear.txt).  Up to four regions, region r of view c = base[r] + c * stride[r] bytes, words[r] ints.
struct ClearJob { unsigned char *base[4]; size_t stride[4]; int64_t words[4]; };
__global__ __launch_bounds__(256) void k_clear_views(ClearJob j)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (!j.base[r]) continue;
        int32_t *q = reinterpret_cast<int32_t *>(j.base[r] + j.stride[r] * blockIdx.y);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j.words[r]; i += (int64_t)gridDim.x * 256) q[i] = 0;
    }
}
void clear_views(hipStream_t s, int C, const ClearJob &j)
{
    int64_t mx = 1;
    for (int r = 0; r < 4; ++r) if (j.base[r] && j.words[r] > mx) mx = j.words[r];
    const unsigned gx = (unsigned)std::min<int64_t>((mx + 1023) / 1024, 256);
    hipLaunchKernelGGL(k_clear_views, dim3(gx, (unsigned)C), dim3(256), 0, s, j);
}

__global__ __launch_bounds__(256) void k_table_scan(int64_t n, const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                    int32_t *sums, int32_t *ticket, VS vs)
{
    __shared__ int wsum[4];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    { const int64_t o = (int64_t)blockIdx.y * vs.ws; in += o; out += o; sums += o; ticket += o; }
    const int64_t base = (int64_t)blockIdx.x * TS_CHUNK + (int64_t)tid * 8;
    int v[8], run = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int64_t i = base + k; run += i < n ? in[i] : 0; v[k] = run; }
    int total;
    const int excl = block_incl_scan256(run, &total, wsum) - run;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int64_t i = base + k; if (i < n) out[i] = v[k] + excl; }
    if (tid == 0) {
        __hip_atomic_store(&sums[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    int carry = 0;
    for (int b0 = 0; b0 < (int)gridDim.x; b0 += 256) {
        const int i = b0 + tid;
        const int x = i < (int)gridDim.x ? __hip_atomic_load(&sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int tot;
        const int incl = block_incl_scan256(x, &tot, wsum);
        if (i < (int)gridDim.x) sums[i] = carry + incl - x;
        carry += tot;
    }
    if (tid == 0) *ticket = 0;         // ready for the next pass
}

// PAIR: keys points at (key, value) uint2 pairs -- the depth passes keep the pair together so that a scattered item is ONE 8-byte store
template <bool PAIR>
__global__ __launch_bounds__(RT) void k_radix_hist(const uint32_t *__restrict__ keys, int64_t n, const int32_t *__restrict__ n_dev,
                                                   int shift, unsigned dmask, int nblocks, int32_t *__restrict__ hist /* [digits][nblocks] */, VS vs,
                                                   int keys_ext /* keys = external [C][vs.ext] pairs, not a workspace buffer */)
{
    { const int64_t o = (int64_t)blockIdx.y * vs.ws; keys += keys_ext ? (int64_t)blockIdx.y * 2 * vs.ext : o; hist += o; if (n_dev) n_dev += blockIdx.y * vs.nd; }
    n = live_count(n, n_dev);
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RB;
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        const int64_t i = base + j * RT + threadIdx.x;
        if (i < n) atomicAdd(&h[((PAIR ? keys[2 * i] : keys[i]) >> shift) & dmask], 1);
    }
    __syncthreads();
    if (threadIdx.x <= dmask) hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// offs = INCLUSIVE scan of hist ([digit][block] flattened); exclusive offset of (d, b) = offs[d*nb + b] - hist[d*nb + b]
template <bool PAIR>
__global__ __launch_bounds__(RT) void k_radix_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                      uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int64_t n,
                                                      const int32_t *__restrict__ n_dev, int shift, int nblocks,
                                                      const int32_t *__restrict__ hist, const int32_t *__restrict__ offs,
                                                      const int32_t *__restrict__ sums, VS vs, int vals_ext, int keys_ext)
{
    {   // view offsets: everything in the per-view workspace region, the last pass's values in the external [C][n] array
        const int64_t o = (int64_t)blockIdx.y * vs.ws;
        keys += keys_ext ? (int64_t)blockIdx.y * 2 * vs.ext : o; hist += o; offs += o; sums += o;
        if (vals) vals += o;
        if (keys_out) keys_out += o;
        if (vals_out) vals_out += vals_ext ? (int64_t)blockIdx.y * vs.ext : o;
        if (n_dev) n_dev += blockIdx.y * vs.nd;
    }
    n = live_count(n, n_dev);
    const int64_t base = (int64_t)blockIdx.x * RB;
    if (base >= n) return;
    // Stable rank with 8 KB of LDS (round 6; rounds 1-5 kept a 64 KB (round, wave) x digit table: two workgroups per CU -- occupancy, not bytes,
    // bounded the pass): wave w owns the CONTIGUOUS items [1024 w, 1024 w + 1024) of the workgroup's 4096 and walks them in 16 rounds of 64, so the
    // order of an item is (wave, round, lane): a per-wave running digit counter, advanced by the leader lane of every digit group of a round
    // (old value broadcast with ds_bpermute), + the ballot rank inside the round + one prefix over the four waves per digit.
    __shared__ int cnt[4 * 256];
    __shared__ int wbase[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 4 * 256; i += RT) cnt[i] = 0;
    __syncthreads();
    uint32_t k[RI], v[RI];
    int pre[RI];
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int *my = cnt + wid * 256;
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        const int64_t i = base + wid * (RI * 64) + j * 64 + lane;
        const bool ok = i < n;
        if (PAIR) {
            const uint2 kv = ok ? reinterpret_cast<const uint2 *>(keys)[i] : make_uint2(0xFFFFFFFFu, 0u);
            k[j] = kv.x; v[j] = kv.y;
        } else {
            k[j] = ok ? keys[i] : 0xFFFFFFFFu;
            v[j] = ok ? vals[i] : 0u;
        }
        const unsigned d = (k[j] >> shift) & 255;
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1);
            m &= ((d >> b) & 1) ? bal : ~bal;
        }
        const bool leader = ok && (m & lt) == 0;
        int old = 0;
        if (leader) { old = my[d]; my[d] = old + __popcll(m); }
        old = __shfl(old, ok ? __builtin_ctzll(m) : lane, 64);
        pre[j] = ok ? old + __popcll(m & lt) : -1;
    }
    __syncthreads();
    {
        const int d = tid;
        const int64_t e = (int64_t)d * nblocks + blockIdx.x;
        int run = offs[e] + sums[e / TS_CHUNK] - hist[e];
#pragma unroll
        for (int w = 0; w < 4; ++w) { wbase[w * 256 + d] = run; run += cnt[w * 256 + d]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        if (pre[j] >= 0) {
            const unsigned d = (k[j] >> shift) & 255;
            const int pos = wbase[wid * 256 + d] + pre[j];
            if (PAIR) {       // keys_out == NULL: last pass, only the values are still needed
                if (keys_out) reinterpret_cast<uint2 *>(keys_out)[pos] = make_uint2(k[j], v[j]);
                else vals_out[pos] = v[j];
            } else {
                keys_out[pos] = k[j];
                vals_out[pos] = v[j];
            }
        }
    }
}

// Scatter of the TILE passes: few digits (2^DB <= 64; the tile-id bits are split evenly over the two passes), so the (round, wave) x digit
// table is small and the 4096 items of the workgroup are first put in digit order in LDS and then streamed out: consecutive lanes write
// consecutive addresses inside each digit run (full-line stores).  The direct form above issues one isolated 4-byte store per item and
// array -- 2 M of them per pass, which is what bounded it (329 us per pass at M = 20 M vs 110 us of bytes at 3 TB/s).
// PK (round 5): the pair travels as ONE word, (tile << idb) | gaussian id (N <= 2^idb, tile bits + idb <= 32): `vals` is not read, the
// non-final pass writes 4 bytes per pair instead of 8; the final pass (vals_out != NULL) writes the packed word (k_tile_bins32 reads the tiles
// from it) AND the plain id into gaussian_ids_sorted.
template <int DB, bool PK>
__global__ __launch_bounds__(RT) void k_radix_scatter_staged(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                             uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int64_t n,
                                                             const int32_t *__restrict__ n_dev, int shift, int nblocks,
                                                             const int32_t *__restrict__ hist, const int32_t *__restrict__ offs,
                                                             const int32_t *__restrict__ sums, VS vs, int vals_ext, uint32_t idmask)
{
    {
        const int64_t o = (int64_t)blockIdx.y * vs.ws;
        keys += o; hist += o; offs += o; sums += o; keys_out += o;
        if (!PK) vals += o;
        if (vals_out) vals_out += vals_ext ? (int64_t)blockIdx.y * vs.ext : o;
        if (n_dev) n_dev += blockIdx.y * vs.nd;
    }
    constexpr int ND = 1 << DB, NS = RI * 4;      // digits, (round, wave) slots
    constexpr int PARTS = RT / ND, SPP = NS / PARTS;   // prefix step: PARTS lanes per digit, SPP slots each
    static_assert(ND <= 64 && NS % PARTS == 0, "digit table");
    __shared__ int tbl[NS * ND];
    __shared__ int part[PARTS * ND];
    __shared__ int lbase[ND], gbase[ND];          // start of the digit's run inside the workgroup's 4096 items / in the output
    __shared__ uint32_t sk[RB], sv[PK ? 1 : RB];
    n = live_count(n, n_dev);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < NS * ND; i += RT) tbl[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RB;
    const int cnt = (int)(n - base < RB ? n - base : RB);
    uint32_t k[RI], v[RI];
    unsigned short rk[RI];
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        const int64_t i = base + j * RT + tid;
        const bool ok = i < n;
        k[j] = ok ? keys[i] : 0xFFFFFFFFu;
        v[j] = (ok && !PK) ? vals[i] : 0u;
        const unsigned d = (k[j] >> shift) & (ND - 1);
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1);
            m &= ((d >> b) & 1) ? bal : ~bal;
        }
        rk[j] = (unsigned short)__popcll(m & lt);
        if (ok && (m & lt) == 0) tbl[(j * 4 + wid) * ND + d] = __popcll(m);
    }
    __syncthreads();
    {   // exclusive prefix over the NS slots of each digit, PARTS lanes per digit
        const int d = tid % ND, pt = tid / ND;
        int run = 0;
#pragma unroll
        for (int q = 0; q < SPP; ++q) {
            const int sidx = (pt * SPP + q) * ND + d;
            const int c = tbl[sidx];
            tbl[sidx] = run;
            run += c;
        }
        part[pt * ND + d] = run;
    }
    __syncthreads();
    if (tid < 64) {   // one wave: per-digit totals -> run starts (wave scan over the digits)
        int tot = 0;
        if (tid < ND)
            for (int pt = 0; pt < PARTS; ++pt) { const int c = part[pt * ND + tid]; part[pt * ND + tid] = tot; tot += c; }
        int inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
        if (tid < ND) {
            lbase[tid] = inc - tot;
            const int64_t e = (int64_t)tid * nblocks + blockIdx.x;
            gbase[tid] = offs[e] + sums[e / TS_CHUNK] - hist[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        if (j * RT + tid < cnt) {
            const unsigned d = (k[j] >> shift) & (ND - 1);
            const int sl = j * 4 + wid;
            const int lp = lbase[d] + part[(sl / SPP) * ND + d] + tbl[sl * ND + d] + rk[j];
            sk[lp] = k[j];
            if (!PK) sv[lp] = v[j];
        }
    }
    __syncthreads();
    for (int i = tid; i < cnt; i += RT) {
        const uint32_t kk = sk[i];
        const unsigned d = (kk >> shift) & (ND - 1);
        const int pos = gbase[d] + (i - lbase[d]);
        keys_out[pos] = kk;
        if (PK) { if (vals_out) vals_out[pos] = kk & idmask; }
        else vals_out[pos] = sv[i];
    }
}

__global__ __launch_bounds__(256) void k_depth_keys(int64_t N, const float *__restrict__ depths, const int32_t *__restrict__ radii,
                                                    uint2 *__restrict__ pairs, VS vs)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    depths += blockIdx.y * vs.src; radii += blockIdx.y * vs.src; pairs += blockIdx.y * (vs.ws / 2);
    pairs[i] = make_uint2(radii[i] > 0 ? __float_as_uint(depths[i]) : 0xFFFFFFFFu, (uint32_t)i);
}

// emission in depth order: pair (tile id, gaussian id) for every tile of the box of order[j].  The 256 Gaussians of a workgroup own ONE
// contiguous output range [cum[j0-1], cum[j0+255]); their hits are laid out in LDS in output order (windows of EW) and streamed out with
// consecutive lanes writing consecutive addresses -- a lane walking its own box would issue isolated 4-byte stores (202 us -> see DESIGN).
constexpr int EW = 4096;
__global__ __launch_bounds__(256) void k_emit_sorted(int64_t N, int64_t M_cap, const uint32_t *__restrict__ order,
                                                     const float *__restrict__ xys, const int32_t *__restrict__ radii,
                                                     const uint32_t *__restrict__ tile_box /* packed boxes instead of (xys, radii), or NULL */,
                                                     const int32_t *__restrict__ cum_sorted, int tiles_x, int tiles_y,
                                                     uint32_t *__restrict__ tile_keys, uint32_t *__restrict__ gids,
                                                     unsigned dmask, int nblocks, int32_t *__restrict__ hist /* [digits][nblocks], zeroed */, VS vs,
                                                     int idb /* > 0: packed pairs, (tile << idb) | id in tile_keys, gids unused */,
                                                     const int32_t *__restrict__ nvis_dev /* != NULL: tile_box is in DEPTH ORDER (box of order[j] at j) and
                                                                                            only the first nvis_dev[view] entries of order / tile_box exist */)
{
    {
        const int64_t o = (int64_t)blockIdx.y * vs.ws, e = (int64_t)blockIdx.y * vs.src;
        order += e; cum_sorted += e; tile_keys += o; gids += o; hist += o;
        if (tile_box) tile_box += e;
        if (xys) { xys += 2 * e; radii += e; }
    }
    const int64_t n_live = nvis_dev ? (int64_t)nvis_dev[blockIdx.y] : N;
    __shared__ uint32_t sT[EW], sG[EW];
    __shared__ int64_t sRange[2];
    __shared__ int sH[2 * 64];                    // first tile pass's digit counts of the (at most two) 4096-blocks a window touches
    const int tid = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * 256, j = j0 + tid;
    if (tid == 0) {
        const int64_t jl = j0 + 255 < N ? j0 + 255 : N - 1;
        sRange[0] = j0 == 0 ? 0 : cum_sorted[j0 - 1];
        sRange[1] = cum_sorted[jl];
    }
    uint32_t g = 0;
    int minx = 0, w = 0, miny = 0;
    int64_t lo = 0, hi = 0;                       // this Gaussian's output range
    if (j < n_live) {
        g = order[j];
        if (tile_box) {            // tight boxes written by gc_project_sh_fwd_boxes (0 = no tile); nvis_dev: read in order, no gather
            const uint32_t bx = tile_box[nvis_dev ? (uint32_t)j : g];
            minx = (int)(bx & 255u); miny = (int)((bx >> 16) & 255u);
            w = (int)((bx >> 8) & 255u) - minx;
            const int hgt = (int)(bx >> 24) - miny;
            if (w > 0 && hgt > 0) {
                lo = j == 0 ? 0 : cum_sorted[j - 1];
                hi = lo + (int64_t)w * hgt;
            } else w = 0;
        }
        const int r = tile_box ? 0 : radii[g];
        if (r > 0) {
            // same float expressions as the projection kernel / oracle (bit-exact tile box)
            const float tcx = xys[2 * (size_t)g] / (float)TILE, tcy = xys[2 * (size_t)g + 1] / (float)TILE, tr = (float)r / (float)TILE;
            minx = clampi((int)(tcx - tr), 0, tiles_x);
            const int maxx = clampi((int)(tcx + tr + 1.f), 0, tiles_x);
            miny = clampi((int)(tcy - tr), 0, tiles_y);
            const int maxy = clampi((int)(tcy + tr + 1.f), 0, tiles_y);
            w = maxx - minx;
            lo = j == 0 ? 0 : cum_sorted[j - 1];
            hi = lo + (int64_t)w * (maxy - miny);
        }
    }
    __syncthreads();
    const int64_t base = sRange[0], total = sRange[1] - base;
    const float rw = w > 0 ? 1.f / (float)w : 0.f;
    for (int64_t win = 0; win < total; win += EW) {
        const int64_t w0 = base + win, w1 = w0 + EW;
        const int64_t a = lo > w0 ? lo : w0, b = hi < w1 ? hi : w1;
        for (int64_t h = a; h < b; ++h) {
            const int i = (int)(h - lo);
            const int q = (int)(((float)i + 0.5f) * rw);            // i / w (exact: i < 2^16 tiles)
            sT[h - w0] = (uint32_t)((miny + q) * tiles_x + minx + (i - q * w));
            sG[h - w0] = g;
        }
        if (tid < 128) sH[tid] = 0;
        __syncthreads();
        const int64_t cnt = total - win < EW ? total - win : EW;
        const int64_t b0 = w0 / RB;
        for (int64_t i = tid; i < cnt; i += 256)
            if (w0 + i < M_cap) {
                const uint32_t t = sT[i];
                if (idb) tile_keys[w0 + i] = (t << idb) | sG[i];
                else { tile_keys[w0 + i] = t; gids[w0 + i] = sG[i]; }
                atomicAdd(&sH[(int)((w0 + i) / RB - b0) * 64 + (int)(t & dmask)], 1);     // saves the pass's k_radix_hist (a read of all M keys)
            }
        __syncthreads();
        if (tid < 128 && sH[tid] != 0 && b0 + (tid >> 6) < nblocks)
            atomicAdd(&hist[(int64_t)(tid & 63) * nblocks + b0 + (tid >> 6)], sH[tid]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_tile_bins32(int64_t M, const int32_t *__restrict__ m_dev, int32_t *__restrict__ overflow,
                                                     int num_tiles, const uint32_t *__restrict__ tkeys,
                                                     const uint32_t *__restrict__ gids, const float *__restrict__ depths,
                                                     int32_t *__restrict__ bins, int64_t *__restrict__ keys64, int32_t *__restrict__ ids_out, VS vs,
                                                     int idb /* > 0: tkeys hold (tile << idb) | id */)
{
    {
        const int64_t o = (int64_t)blockIdx.y * vs.ws;
        tkeys += o; bins += (int64_t)blockIdx.y * 2 * num_tiles; if (depths) depths += blockIdx.y * vs.src;
        gids += (int64_t)blockIdx.y * vs.ext;             // (the last radix pass wrote the ids into the external [C][M_cap] array)
        if (keys64) keys64 += (int64_t)blockIdx.y * vs.ext;
        if (m_dev) { m_dev += blockIdx.y * vs.nd; overflow += blockIdx.y * vs.nd; }
    }
    if (m_dev && overflow && blockIdx.x == 0 && threadIdx.x == 0) *overflow = (int64_t)*m_dev > M ? 1 : 0;
    M = live_count(M, m_dev);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int t = (int)(tkeys[i] >> idb);
    if (ids_out || keys64) {            // the last radix pass normally writes the ids straight into gaussian_ids_sorted (ids_out == NULL)
        const uint32_t g = gids[i];
        if (ids_out) ids_out[i] = (int32_t)g;
        if (keys64) keys64[i] = ((int64_t)t << 32) | (int64_t)__float_as_uint(depths[g]);
    }
    if (i == 0) bins[2 * t] = 0;
    else {
        const int tp = (int)(tkeys[i - 1] >> idb);
        if (tp != t) { bins[2 * tp + 1] = (int32_t)i; bins[2 * t] = (int32_t)i; }
    }
    if (i == M - 1) bins[2 * t + 1] = (int32_t)M;
}

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace of a run of radix passes over n (key32, val32) pairs: 2 ping-pong pair buffers + digit tables + scan scratch
struct Plan {
    size_t off_keys[2], off_vals[2], off_hist, off_offs, off_scan, scan_bytes, off_cnt, total;
    int nb;
};

Plan make_plan(int64_t n)
{
    Plan p;
    p.nb = (int)((n + RB - 1) / RB);
    if (p.nb < 1) p.nb = 1;
    size_t o = 0;
    for (int i = 0; i < 2; ++i) { p.off_keys[i] = o; o += al(4 * (size_t)n + 4); p.off_vals[i] = o; o += al(4 * (size_t)n + 4); }
    p.off_hist = o; o += al(4 * 256 * (size_t)p.nb);
    p.off_offs = o; o += al(4 * 256 * (size_t)p.nb);
    const int64_t scan_n = 256 * (int64_t)p.nb > n ? 256 * (int64_t)p.nb : n;
    p.scan_bytes = al(gc_raster_scan_workspace_bytes(scan_n));
    p.off_scan = o; o += p.scan_bytes;
    p.off_cnt = o; o += 256;
    p.total = o;
    return p;
}

void set_attr() {}

// one stable radix pass of `dbits` bits (8: the direct scatter; 5 / 6: the LDS-staged scatter of the tile passes) over C views at once
// (grid.y = view; per-view workspace regions of vs.ws words).  vals_ext: `vo` is the external [C][vs.ext] array, not a workspace buffer.
int radix_pass(const uint32_t *ki, const uint32_t *vi, uint32_t *ko, uint32_t *vo, int64_t n, const int32_t *n_dev, int shift,
               const Plan &p, unsigned char *w, hipStream_t s, int C, const VS &vs, int vals_ext, int dbits = 8, bool pair = false,
               bool have_hist = false, int keys_ext = 0, int idb = 0)
{
    int32_t *hist = (int32_t *)(w + p.off_hist), *offs = (int32_t *)(w + p.off_offs), *cnt = (int32_t *)(w + p.off_cnt);
    const int nd = 1 << dbits;
    const dim3 g((unsigned)p.nb, (unsigned)C);
    if (have_hist) {}                     // the producer of `ki` already accumulated this pass's histogram (k_emit_sorted)
    else if (pair) hipLaunchKernelGGL(k_radix_hist<true>, g, dim3(RT), 0, s, ki, n, n_dev, shift, (unsigned)(nd - 1), p.nb, hist, vs, keys_ext);
    else hipLaunchKernelGGL(k_radix_hist<false>, g, dim3(RT), 0, s, ki, n, n_dev, shift, (unsigned)(nd - 1), p.nb, hist, vs, 0);
    // cnt[0] is the scan's ticket (zeroed once per phase, self-resetting); the scan scratch holds the per-2048-entry offsets
    int32_t *sums = (int32_t *)(w + p.off_scan);
    const int64_t ne = nd * (int64_t)p.nb;
    hipLaunchKernelGGL(k_table_scan, dim3((unsigned)((ne + TS_CHUNK - 1) / TS_CHUNK), (unsigned)C), dim3(256), 0, s, ne, hist, offs, sums, cnt, vs);
    const uint32_t idmask = idb ? ((1u << idb) - 1u) : 0u;
    if (dbits == 5 && idb)
        hipLaunchKernelGGL((k_radix_scatter_staged<5, true>), g, dim3(RT), 0, s, ki, vi, ko, vo, n, n_dev, shift, p.nb, hist, offs, sums, vs, vals_ext, idmask);
    else if (dbits == 6 && idb)
        hipLaunchKernelGGL((k_radix_scatter_staged<6, true>), g, dim3(RT), 0, s, ki, vi, ko, vo, n, n_dev, shift, p.nb, hist, offs, sums, vs, vals_ext, idmask);
    else if (dbits == 5)
        hipLaunchKernelGGL((k_radix_scatter_staged<5, false>), g, dim3(RT), 0, s, ki, vi, ko, vo, n, n_dev, shift, p.nb, hist, offs, sums, vs, vals_ext, idmask);
    else if (dbits == 6)
        hipLaunchKernelGGL((k_radix_scatter_staged<6, false>), g, dim3(RT), 0, s, ki, vi, ko, vo, n, n_dev, shift, p.nb, hist, offs, sums, vs, vals_ext, idmask);
    else if (pair)
        hipLaunchKernelGGL(k_radix_scatter<true>, g, dim3(RT), 0, s, ki, vi, ko, vo, n, n_dev,
                           shift, p.nb, hist, offs, sums, vs, vals_ext, keys_ext);
    else
        hipLaunchKernelGGL(k_radix_scatter<false>, g, dim3(RT), 0, s, ki, vi, ko, vo, n, n_dev,
                           shift, p.nb, hist, offs, sums, vs, vals_ext, 0);
    return GC_OK;
}

// per-view region of the depth-order workspace: the radix plan + nothing else (the gathered tile counts are no longer materialised)
size_t depth_region(int64_t N) { return make_plan(N > 0 ? N : 1).total; }

int depth_order_impl(int64_t N, int C, const float *depths, const int32_t *radii, const uint32_t *pairs_in, const int32_t *num_tiles_hit,
                     int32_t *depth_order, int32_t *cum_sorted, int32_t *count_dev, void *workspace, size_t workspace_bytes,
                     void *stream, const char *what)
{
    hipStream_t s = gc::S(stream);
    if (N == 0) return hipMemsetAsync(count_dev, 0, 4 * (size_t)C, s) == hipSuccess ? GC_OK : GC_ELAUNCH;
    if (!((pairs_in || (depths && radii)) && num_tiles_hit && depth_order && cum_sorted && workspace)) { gc::set_error("%s: null pointer", what); return GC_EINVAL; }
    const Plan p = make_plan(N);
    const size_t region = depth_region(N);
    if (workspace_bytes < region * (size_t)C) { gc::set_error("%s: workspace too small", what); return GC_ENOSPC; }
    set_attr();
    unsigned char *w = (unsigned char *)workspace;
    VS vs; vs.ws = (int64_t)(region / 4); vs.ext = N; vs.src = N; vs.nd = 0;
    uint32_t *k0 = (uint32_t *)(w + p.off_keys[0]), *k1 = (uint32_t *)(w + p.off_keys[1]);
    // (key, id) pairs ping-pong between the two halves of the view's region (each half = the keys + vals regions of the plan, >= 8 N bytes)
    uint32_t *pa = k0, *pb = k1;
    { ClearJob cj = {}; cj.base[0] = w + p.off_cnt; cj.stride[0] = region; cj.words[0] = 1; clear_views(s, C, cj); }     // tickets of k_table_scan
    // pairs_in: the projection kernel already wrote the (depth bits | 0xFFFFFFFF, id) pairs ([C][N] uint2): the first pass reads them in place
    if (!pairs_in)
        hipLaunchKernelGGL(k_depth_keys, dim3(gc::cdiv(N, 256), (unsigned)C), dim3(256), 0, s, N, depths, radii, (uint2 *)pa, vs);
    for (int pass = 0; pass < 4; ++pass) {   // visible depths are > 0: the float bit pattern is monotone
        const bool ext = pass == 0 && pairs_in;
        int rc = radix_pass(ext ? pairs_in : pa, nullptr, pass == 3 ? nullptr : pb, (uint32_t *)depth_order, N, nullptr, 8 * pass, p, w, s, C, vs, 1,
                            8, true, false, ext ? 1 : 0);
        if (rc != GC_OK) return rc;
        uint32_t *t = pa; pa = pb; pb = t;
    }
    // inclusive scan of the tile counts TAKEN IN DEPTH ORDER (the gather is fused into the scan's load), per view
    int rc = gc_raster_scan_tiles_views(N, C, num_tiles_hit, depth_order, cum_sorted, count_dev, w + p.off_scan, p.scan_bytes,
                                        (int64_t)region, (void *)s);
    if (rc != GC_OK) return rc;
    return gc::check_launch(what);
}

}  // namespace

extern "C" {

size_t gc_raster_depth_order_workspace_bytes(int64_t N) { return depth_region(N); }
size_t gc_raster_depth_order_views_workspace_bytes(int64_t N, int C) { return depth_region(N) * (size_t)(C > 0 ? C : 1); }

/* Phase 1 (no host sync): depth_order[N] = Gaussian ids sorted by (depth bits, id), culled (radii <= 0) last;
 * cum_sorted[N] = inclusive scan of num_tiles_hit taken in that order; *count_dev = M. */
int gc_raster_depth_order(int64_t N, const float *depths, const int32_t *radii, const int32_t *num_tiles_hit,
                          int32_t *depth_order, int32_t *cum_sorted, int32_t *count_dev, void *workspace, size_t workspace_bytes,
                          void *stream)
{
    GC_REQUIRE(N >= 0 && count_dev, "bad arguments");
    return depth_order_impl(N, 1, depths, radii, nullptr, num_tiles_hit, depth_order, cum_sorted, count_dev, workspace, workspace_bytes, stream,
                            "gc_raster_depth_order");
}

/* The same for C views in ONE set of launches (grid.y = view): depths / radii / num_tiles_hit / depth_order / cum_sorted are [C][N],
 * count_dev[C].  depth_pairs (optional, [C][N] uint2 = (depth bits or 0xFFFFFFFF when culled, Gaussian id), as gc_project_sh_fwd_views
 * writes them) replaces the (depths, radii) read. */
int gc_raster_depth_order_views(int64_t N, int C, const float *depths, const int32_t *radii, const uint32_t *depth_pairs,
                                const int32_t *num_tiles_hit, int32_t *depth_order, int32_t *cum_sorted, int32_t *count_dev,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    GC_REQUIRE(N >= 0 && C >= 1 && C <= 65535 && count_dev, "bad arguments");
    return depth_order_impl(N, C, depths, radii, depth_pairs, num_tiles_hit, depth_order, cum_sorted, count_dev, workspace, workspace_bytes,
                            stream, "gc_raster_depth_order_views");
}

size_t gc_raster_bin_workspace_bytes(int64_t M) { return make_plan(M > 0 ? M : 1).total; }
size_t gc_raster_bin_views_workspace_bytes(int64_t M_cap, int C) { return make_plan(M_cap > 0 ? M_cap : 1).total * (size_t)(C > 0 ? C : 1); }

}  // extern "C"

namespace {
int bin_tiles_impl(int64_t N, int C, int64_t M, const int32_t *m_dev, int32_t *overflow_dev, const int32_t *depth_order,
                   const int32_t *cum_sorted, const float *xys, const float *depths, const int32_t *radii, const uint32_t *tile_boxes,
                   int tiles_x, int tiles_y,
                   int32_t *gaussian_ids_sorted, int32_t *tile_bins, int64_t *isect_ids_sorted, void *workspace,
                   size_t workspace_bytes, void *stream, const char *what, const int32_t *visible_dev = nullptr)
{
    const int num_tiles = tiles_x * tiles_y;
    hipStream_t s = gc::S(stream);
    if (M == 0 || N == 0 || !workspace) {      // nothing to bin: empty lists (the general path clears these together with its counters, below)
        if (hipMemsetAsync(tile_bins, 0, sizeof(int32_t) * 2 * (size_t)num_tiles * C, s) != hipSuccess) return GC_ELAUNCH;
        if (overflow_dev && hipMemsetAsync(overflow_dev, 0, 4 * (size_t)C, s) != hipSuccess) return GC_ELAUNCH;
        if (M == 0 || N == 0) return GC_OK;
    }
    if (num_tiles > 65536) { gc::set_error("%s: at most 65536 tiles", what); return GC_EINVAL; }
    if (!(depth_order && cum_sorted && ((xys && radii) || tile_boxes) && (depths || !isect_ids_sorted) && gaussian_ids_sorted && workspace)) { gc::set_error("%s: null pointer", what); return GC_EINVAL; }
    const Plan p = make_plan(M);
    if (workspace_bytes < p.total * (size_t)C) { gc::set_error("%s: workspace too small", what); return GC_ENOSPC; }
    set_attr();
    unsigned char *w = (unsigned char *)workspace;
    VS vs; vs.ws = (int64_t)(p.total / 4); vs.ext = M; vs.src = N; vs.nd = m_dev ? 1 : 0;
    uint32_t *k0 = (uint32_t *)(w + p.off_keys[0]), *v0 = (uint32_t *)(w + p.off_vals[0]);
    uint32_t *k1 = (uint32_t *)(w + p.off_keys[1]), *v1 = (uint32_t *)(w + p.off_vals[1]);
    // the tile-id bits are split evenly over two passes (1024 tiles: 5 + 5, 4096: 6 + 6) and scattered through LDS; above 12 bits
    // (or for a single pass) the 8-bit direct scatter runs
    int tbits = 1;
    while ((1 << tbits) < num_tiles) ++tbits;
    const int passes = tbits <= 6 ? 1 : 2;
    const int dbits = tbits <= 5 ? 5 : (tbits <= 6 ? 6 : (tbits <= 10 ? 5 : (tbits <= 12 ? 6 : 8)));
    const bool fused_hist = dbits <= 6;          // the emission also counts the first pass's digits (64 LDS counters per 4096-block)
    {   // tile_bins, overflow flags, the tickets of k_table_scan and the first pass's digit tables (counted by the emission) of every view: one launch
        ClearJob cj = {};
        cj.base[0] = (unsigned char *)tile_bins; cj.stride[0] = sizeof(int32_t) * 2 * (size_t)num_tiles; cj.words[0] = 2 * (int64_t)num_tiles;
        if (overflow_dev) { cj.base[1] = (unsigned char *)overflow_dev; cj.stride[1] = 4; cj.words[1] = 1; }
        cj.base[2] = w + p.off_cnt; cj.stride[2] = p.total; cj.words[2] = 1;
        if (fused_hist) { cj.base[3] = w + p.off_hist; cj.stride[3] = p.total; cj.words[3] = ((int64_t)1 << dbits) * p.nb; }
        clear_views(s, C, cj);
    }
    // packed pairs: (tile << idb) | id in ONE word when the id and tile bits fit 32 and the staged (5 / 6-bit) scatter runs -- 1 024 tiles and
    // N <= 4 M: every pass moves 4 bytes per pair instead of 8 (emit 4, pass 1 4 + 4, pass 2 4 + 8, bins 4 = 28 instead of 44 bytes per pair)
    int idb = 1;
    while (((int64_t)1 << idb) < N) ++idb;
    if (!(dbits <= 6 && passes * dbits + idb <= 32)) idb = 0;
    hipLaunchKernelGGL(k_emit_sorted, dim3(gc::cdiv(N, 256), (unsigned)C), dim3(256), 0, s, N, M, (const uint32_t *)depth_order, xys, radii,
                       tile_boxes, cum_sorted, tiles_x, tiles_y, k0, v0, fused_hist ? (1u << dbits) - 1u : 0u, fused_hist ? p.nb : 0,
                       (int32_t *)(w + p.off_hist), vs, idb, visible_dev);
    uint32_t *ks = k0, *vsrc = v0;
    for (int pass = 0; pass < passes; ++pass) {
        const bool last = pass == passes - 1;
        uint32_t *ko = ks == k0 ? k1 : k0, *vo = last ? (uint32_t *)gaussian_ids_sorted : (idb ? (uint32_t *)nullptr : (vsrc == v0 ? v1 : v0));
        int rc = radix_pass(ks, vsrc, ko, vo, M, m_dev, idb + dbits * pass, p, w, s, C, vs, last ? 1 : 0, dbits, false, fused_hist && pass == 0, 0, idb);
        if (rc != GC_OK) return rc;
        ks = ko; vsrc = vo;
    }
    hipLaunchKernelGGL(k_tile_bins32, dim3(gc::cdiv(M, 256), (unsigned)C), dim3(256), 0, s, M, m_dev, overflow_dev, num_tiles, ks,
                       (const uint32_t *)gaussian_ids_sorted, depths, tile_bins, isect_ids_sorted, (int32_t *)nullptr, vs, idb);
    return gc::check_launch(what);
}
}  // namespace

extern "C" {

/* Phase 2: emit (tile, id) pairs in depth order for the M intersections, stable-sort them by tile, build tile_bins.
 * gaussian_ids_sorted[M] and tile_bins[T,2] are the rasterizer's inputs; isect_ids_sorted[M] (int64 tile<<32|depth bits) is
 * optional (NULL to skip). */
int gc_raster_bin_tiles(int64_t N, int64_t M, const int32_t *depth_order, const int32_t *cum_sorted, const float *xys,
                        const float *depths, const int32_t *radii, int tiles_x, int tiles_y, int32_t *gaussian_ids_sorted,
                        int32_t *tile_bins, int64_t *isect_ids_sorted, void *workspace, size_t workspace_bytes, void *stream)
{
    GC_REQUIRE(N >= 0 && M >= 0 && tile_bins, "bad arguments");
    return bin_tiles_impl(N, 1, M, nullptr, nullptr, depth_order, cum_sorted, xys, depths, radii, nullptr, tiles_x, tiles_y, gaussian_ids_sorted,
                          tile_bins, isect_ids_sorted, workspace, workspace_bytes, stream, "gc_raster_bin_tiles");
}

/* Sync-free phase 2: the intersection count stays on the device (count_dev, written by gc_raster_depth_order); every buffer
 * is sized for the caller's capacity M_cap (gaussian_ids_sorted[M_cap], workspace gc_raster_bin_workspace_bytes(M_cap)), the
 * kernels clamp to min(*count_dev, M_cap) and *overflow_dev is set to 1 when the frame needed more (the image is then
 * incomplete: re-run with a larger capacity).  No host round trip: the blocking gc_raster_read_count is not needed. */
int gc_raster_bin_tiles_dev(int64_t N, int64_t M_cap, const int32_t *count_dev, int32_t *overflow_dev, const int32_t *depth_order,
                            const int32_t *cum_sorted, const float *xys, const float *depths, const int32_t *radii, int tiles_x,
                            int tiles_y, int32_t *gaussian_ids_sorted, int32_t *tile_bins, int64_t *isect_ids_sorted,
                            void *workspace, size_t workspace_bytes, void *stream)
{
    GC_REQUIRE(N >= 0 && M_cap >= 0 && tile_bins && count_dev && overflow_dev, "bad arguments");
    return bin_tiles_impl(N, 1, M_cap, count_dev, overflow_dev, depth_order, cum_sorted, xys, depths, radii, nullptr, tiles_x, tiles_y,
                          gaussian_ids_sorted, tile_bins, isect_ids_sorted, workspace, workspace_bytes, stream, "gc_raster_bin_tiles_dev");
}

/* Phase 2 on the tight tile boxes of gc_project_sh_fwd_boxes (instead of the boxes recomputed from xys / radii).  count_dev == NULL:
 * M is the exact intersection count (the form of gc_raster_bin_tiles); otherwise the sync-free form of gc_raster_bin_tiles_dev
 * (M = capacity, overflow_dev required). */
int gc_raster_bin_tiles_boxes(int64_t N, int64_t M, const int32_t *count_dev, int32_t *overflow_dev, const int32_t *depth_order,
                              const int32_t *cum_sorted, const uint32_t *tile_boxes, const float *depths, int tiles_x, int tiles_y,
                              int32_t *gaussian_ids_sorted, int32_t *tile_bins, int64_t *isect_ids_sorted, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    GC_REQUIRE(N >= 0 && M >= 0 && tile_bins && tile_boxes && ((count_dev == nullptr) == (overflow_dev == nullptr)), "bad arguments");
    GC_REQUIRE(tiles_x <= 255 && tiles_y <= 255, "packed boxes hold at most 255 x 255 tiles");
    return bin_tiles_impl(N, 1, M, count_dev, overflow_dev, depth_order, cum_sorted, nullptr, depths, nullptr, tile_boxes, tiles_x, tiles_y,
                          gaussian_ids_sorted, tile_bins, isect_ids_sorted, workspace, workspace_bytes, stream, "gc_raster_bin_tiles_boxes");
}

/* Phase 2 for C views in one set of launches (always sync-free, always on the packed tight boxes): count_dev[C] / overflow_dev[C],
 * depth_order / cum_sorted / tile_boxes / depths [C][N], gaussian_ids_sorted [C][M_cap], tile_bins [C][T][2], isect_ids_sorted [C][M_cap]
 * or NULL; workspace >= gc_raster_bin_views_workspace_bytes(M_cap, C). */
int gc_raster_bin_tiles_views(int64_t N, int C, int64_t M_cap, const int32_t *count_dev, int32_t *overflow_dev, const int32_t *depth_order,
                              const int32_t *cum_sorted, const uint32_t *tile_boxes, const float *depths, int tiles_x, int tiles_y,
                              int32_t *gaussian_ids_sorted, int32_t *tile_bins, int64_t *isect_ids_sorted, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    GC_REQUIRE(N >= 0 && C >= 1 && C <= 65535 && M_cap >= 0 && tile_bins && tile_boxes && count_dev && overflow_dev, "bad arguments");
    GC_REQUIRE(tiles_x <= 255 && tiles_y <= 255, "packed boxes hold at most 255 x 255 tiles");
    return bin_tiles_impl(N, C, M_cap, count_dev, overflow_dev, depth_order, cum_sorted, nullptr, depths, nullptr, tile_boxes, tiles_x, tiles_y,
                          gaussian_ids_sorted, tile_bins, isect_ids_sorted, workspace, workspace_bytes, stream, "gc_raster_bin_tiles_views");
}

}  // extern "C"


// ================================================================================================================================
// Round 6: depth order WITHOUT the per-Gaussian gathers.  gc_raster_depth_order_views sorts (depth bits, id) pairs; afterwards the scan reads
// num_tiles_hit[order[j]] and the emission tile_boxes[order[j]] -- two random 4-byte reads per Gaussian, i.e. two 128-byte line fetches:
// by the counters (profiles/r05_raster_traffic_views8.json) 2/3 of everything the depth order fetched.  Here the packed tight box rides through
// the radix passes as a third word of the item (12-byte (key, id, box) triples), so the scan (count = box area) and the emission read their
// inputs sequentially; and the Gaussians the projection culled (key 0xFFFFFFFF) are dropped by the FIRST pass instead of travelling through
// all four (its rank only counts visible items; the later passes run on visible_dev[view] items).
// The order is the same stable LSD sort by (depth bits, id) over the same visible set: gaussian_ids_sorted / tile_bins are bit-identical.
namespace {

struct __attribute__((packed, aligned(4))) Tri { uint32_t k, id, box; };

// P0: the items are the projection's external (key, id) pairs [C][N] + tile boxes [C][N], culled items (key 0xFFFFFFFF) do not count
template <bool P0>
__global__ __launch_bounds__(RT) void k_tri_hist(const void *__restrict__ items, int64_t n, const int32_t *__restrict__ n_dev, int shift,
                                                 int nblocks, int32_t *__restrict__ hist, int32_t *__restrict__ visible_dev, VS vs)
{
    hist += (int64_t)blockIdx.y * vs.ws;
    const uint32_t *keys;
    if (P0) keys = (const uint32_t *)items + (int64_t)blockIdx.y * 2 * vs.ext;
    else { keys = (const uint32_t *)items + (int64_t)blockIdx.y * vs.ws; n = (int64_t)n_dev[blockIdx.y] < n ? (int64_t)n_dev[blockIdx.y] : n; }
    __shared__ int h[256];
    __shared__ int vis;
    h[threadIdx.x] = 0;
    if (threadIdx.x == 0) vis = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RB;
    int mine = 0;
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        const int64_t i = base + j * RT + threadIdx.x;
        if (i < n) {
            const uint32_t k = keys[(P0 ? 2 : 3) * i];
            if (!P0 || k != 0xFFFFFFFFu) { atomicAdd(&h[(k >> shift) & 255], 1); ++mine; }
        }
    }
    if (P0 && mine) atomicAdd(&vis, mine);
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
    if (P0 && threadIdx.x == 0 && vis) atomicAdd(&visible_dev[blockIdx.y], vis);
}

// LAST: the sorted ids and boxes go to the external [C][N] arrays (the keys are not needed any more).
// Stable rank with 8 KB of LDS (the pair kernel above keeps a 64 KB (round, wave) x digit table: two workgroups per CU, and by the counters
// its passes run at a quarter of the HBM rate although they move few bytes -- occupancy, not traffic, bounds them).  Here wave w owns the
// CONTIGUOUS items [1024 w, 1024 w + 1024) of the workgroup's 4096 and walks them in 16 rounds of 64: the order of an item is (wave, round,
// lane), so a per-wave running digit counter (cnt[w][d], read and advanced by the leader lane of every digit group of a round, the old value
// broadcast to the group with ds_bpermute) gives the rank among the wave's earlier items, the ballots the rank inside the round, and one
// prefix over the four waves per digit (thread d) the rest.  8+ workgroups per CU hide the round-to-round LDS dependency.
template <bool P0, bool LAST>
__global__ __launch_bounds__(RT) void k_tri_scatter(const void *__restrict__ items, const uint32_t *__restrict__ boxes_in, Tri *__restrict__ out,
                                                    uint32_t *__restrict__ ids_out, uint32_t *__restrict__ boxes_out, int64_t n,
                                                    const int32_t *__restrict__ n_dev, int shift, int nblocks, const int32_t *__restrict__ hist,
                                                    const int32_t *__restrict__ offs, const int32_t *__restrict__ sums, VS vs)
{
    {
        const int64_t o = (int64_t)blockIdx.y * vs.ws;
        hist += o; offs += o; sums += o;
        if (!LAST) out = (Tri *)((uint32_t *)out + o);
        else { ids_out += (int64_t)blockIdx.y * vs.ext; boxes_out += (int64_t)blockIdx.y * vs.ext; }
        if (P0) boxes_in += (int64_t)blockIdx.y * vs.ext;
    }
    const uint32_t *src = P0 ? (const uint32_t *)items + (int64_t)blockIdx.y * 2 * vs.ext : (const uint32_t *)items + (int64_t)blockIdx.y * vs.ws;
    if (!P0) n = (int64_t)n_dev[blockIdx.y] < n ? (int64_t)n_dev[blockIdx.y] : n;
    const int64_t base = (int64_t)blockIdx.x * RB;
    if (base >= n) return;
    __shared__ int cnt[4 * 256];      // running count of digit d among the items wave w has walked
    __shared__ int wbase[4 * 256];    // output position of wave w's first item with digit d
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 4 * 256; i += RT) cnt[i] = 0;
    __syncthreads();
    uint32_t k[RI], v[RI], bx[RI];
    int pre[RI];                      // rank among the wave's items with the same digit (-1: no item)
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int *my = cnt + wid * 256;
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        const int64_t i = base + wid * (RI * 64) + j * 64 + lane;
        bool ok = i < n;
        k[j] = 0xFFFFFFFFu; v[j] = 0u; bx[j] = 0u;
        if (ok) {
            if (P0) {
                const uint2 kv = reinterpret_cast<const uint2 *>(src)[i];
                k[j] = kv.x; v[j] = kv.y;
                ok = kv.x != 0xFFFFFFFFu;
                if (ok) bx[j] = boxes_in[i];
            } else {
                const Tri t = reinterpret_cast<const Tri *>(src)[i];
                k[j] = t.k; v[j] = t.id; bx[j] = t.box;
            }
        }
        const unsigned d = (k[j] >> shift) & 255;
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1);
            m &= ((d >> b) & 1) ? bal : ~bal;
        }
        const bool leader = ok && (m & lt) == 0;
        int old = 0;
        if (leader) { old = my[d]; my[d] = old + __popcll(m); }        // (LDS operations of one wave execute in program order: round j + 1 sees this)
        const int lead_lane = ok ? __builtin_ctzll(m) : lane;
        old = __shfl(old, lead_lane, 64);
        pre[j] = ok ? old + __popcll(m & lt) : -1;
    }
    __syncthreads();
    {   // thread d: where each wave's items of digit d start (global offset of (d, this workgroup) + the earlier waves' totals)
        const int d = tid;
        const int64_t e = (int64_t)d * nblocks + blockIdx.x;
        int run = offs[e] + sums[e / TS_CHUNK] - hist[e];
#pragma unroll
        for (int w = 0; w < 4; ++w) { wbase[w * 256 + d] = run; run += cnt[w * 256 + d]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        if (pre[j] >= 0) {
            const unsigned d = (k[j] >> shift) & 255;
            const int pos = wbase[wid * 256 + d] + pre[j];
            if (LAST) { ids_out[pos] = v[j]; boxes_out[pos] = bx[j]; }
            else { Tri t; t.k = k[j]; t.id = v[j]; t.box = bx[j]; out[pos] = t; }
        }
    }
}

// tile count of the box at depth position j (0 past the visible ones), written where the in-place scan turns it into cum_sorted
__global__ __launch_bounds__(256) void k_box_counts(int64_t N, const uint32_t *__restrict__ boxes_sorted, const int32_t *__restrict__ visible_dev,
                                                    int32_t *__restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    boxes_sorted += blockIdx.y * N; cnt += blockIdx.y * N;
    int c = 0;
    if (i < (int64_t)visible_dev[blockIdx.y]) {
        const uint32_t bx = boxes_sorted[i];
        const int w = (int)((bx >> 8) & 255u) - (int)(bx & 255u), h = (int)(bx >> 24) - (int)((bx >> 16) & 255u);
        c = (w > 0 && h > 0) ? w * h : 0;
    }
    cnt[i] = c;
}

// per-view workspace of the triple sort: 2 ping-pong triple buffers + digit tables + scan scratch + ticket
struct TriPlan { size_t off_buf[2], off_hist, off_offs, off_scan, scan_bytes, off_cnt, total; int nb; };
TriPlan make_tri_plan(int64_t n)
{
    TriPlan p;
    p.nb = (int)((n + RB - 1) / RB);
    if (p.nb < 1) p.nb = 1;
    size_t o = 0;
    for (int i = 0; i < 2; ++i) { p.off_buf[i] = o; o += al(12 * (size_t)n + 16); }
    p.off_hist = o; o += al(4 * 256 * (size_t)p.nb);
    p.off_offs = o; o += al(4 * 256 * (size_t)p.nb);
    const int64_t scan_n = 256 * (int64_t)p.nb > n ? 256 * (int64_t)p.nb : n;
    p.scan_bytes = al(gc_raster_scan_workspace_bytes(scan_n));
    p.off_scan = o; o += p.scan_bytes;
    p.off_cnt = o; o += 256;
    p.total = o;
    return p;
}

void set_attr_tri() {}

}  // namespace

extern "C" {

size_t gc_raster_order_boxes_views_workspace_bytes(int64_t N, int C) { return make_tri_plan(N > 0 ? N : 1).total * (size_t)(C > 0 ? C : 1); }

/* Depth order of C views on the packed tight boxes (see the block comment above): depth_pairs [C][N][2] and tile_boxes [C][N] as
 * gc_project_sh_fwd_views writes them -> depth_order [C][N] = ids of the VISIBLE Gaussians in (depth bits, id) order (entries past
 * visible_dev[c] are unspecified), boxes_sorted [C][N] = their boxes in that order, cum_sorted [C][N] = inclusive scan of the boxes' tile counts
 * (constant past the visible ones), count_dev [C] = M, visible_dev [C]. */
int gc_raster_order_boxes_views(int64_t N, int C, const uint32_t *depth_pairs, const uint32_t *tile_boxes, int32_t *depth_order,
                                uint32_t *boxes_sorted, int32_t *cum_sorted, int32_t *count_dev, int32_t *visible_dev, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    GC_REQUIRE(N >= 0 && C >= 1 && C <= 65535 && count_dev && visible_dev, "bad arguments");
    hipStream_t s = gc::S(stream);
    if (N == 0) {
        if (hipMemsetAsync(visible_dev, 0, 4 * (size_t)C, s) != hipSuccess) return GC_ELAUNCH;
        return hipMemsetAsync(count_dev, 0, 4 * (size_t)C, s) == hipSuccess ? GC_OK : GC_ELAUNCH;
    }
    GC_REQUIRE(depth_pairs && tile_boxes && depth_order && boxes_sorted && cum_sorted && workspace, "null pointer");
    const TriPlan p = make_tri_plan(N);
    if (workspace_bytes < p.total * (size_t)C) { gc::set_error("gc_raster_order_boxes_views: workspace too small"); return GC_ENOSPC; }
    set_attr_tri();
    unsigned char *w = (unsigned char *)workspace;
    VS vs; vs.ws = (int64_t)(p.total / 4); vs.ext = N; vs.src = N; vs.nd = 1;
    {   // visible counts + the tickets of k_table_scan: one launch
        ClearJob cj = {};
        cj.base[0] = (unsigned char *)visible_dev; cj.stride[0] = 4; cj.words[0] = 1;
        cj.base[1] = w + p.total * (size_t)0 + p.off_cnt; cj.stride[1] = p.total; cj.words[1] = 1;
        clear_views(s, C, cj);
    }
    int32_t *hist = (int32_t *)(w + p.off_hist), *offs = (int32_t *)(w + p.off_offs), *tick = (int32_t *)(w + p.off_cnt), *sums = (int32_t *)(w + p.off_scan);
    const dim3 g((unsigned)p.nb, (unsigned)C);
    const int64_t ne = 256 * (int64_t)p.nb;
    const dim3 gscan((unsigned)((ne + TS_CHUNK - 1) / TS_CHUNK), (unsigned)C);
    const size_t lds = 0;
    Tri *a = (Tri *)(w + p.off_buf[0]), *b = (Tri *)(w + p.off_buf[1]);
    // pass 0: external pairs + boxes in, visible triples out
    hipLaunchKernelGGL(k_tri_hist<true>, g, dim3(RT), 0, s, (const void *)depth_pairs, N, (const int32_t *)nullptr, 0, p.nb, hist, visible_dev, vs);
    hipLaunchKernelGGL(k_table_scan, gscan, dim3(256), 0, s, ne, hist, offs, sums, tick, vs);
    hipLaunchKernelGGL((k_tri_scatter<true, false>), g, dim3(RT), lds, s, (const void *)depth_pairs, tile_boxes, a, (uint32_t *)nullptr, (uint32_t *)nullptr, N,
                       (const int32_t *)nullptr, 0, p.nb, hist, offs, sums, vs);
    for (int pass = 1; pass < 4; ++pass) {   // visible depths are > 0: the float bit pattern is monotone
        hipLaunchKernelGGL(k_tri_hist<false>, g, dim3(RT), 0, s, (const void *)a, N, (const int32_t *)visible_dev, 8 * pass, p.nb, hist, (int32_t *)nullptr, vs);
        hipLaunchKernelGGL(k_table_scan, gscan, dim3(256), 0, s, ne, hist, offs, sums, tick, vs);
        if (pass < 3)
            hipLaunchKernelGGL((k_tri_scatter<false, false>), g, dim3(RT), lds, s, (const void *)a, (const uint32_t *)nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr,
                               N, (const int32_t *)visible_dev, 8 * pass, p.nb, hist, offs, sums, vs);
        else
            hipLaunchKernelGGL((k_tri_scatter<false, true>), g, dim3(RT), lds, s, (const void *)a, (const uint32_t *)nullptr, (Tri *)nullptr, (uint32_t *)depth_order, boxes_sorted,
                               N, (const int32_t *)visible_dev, 8 * pass, p.nb, hist, offs, sums, vs);
        Tri *t = a; a = b; b = t;
    }
    hipLaunchKernelGGL(k_box_counts, dim3(gc::cdiv(N, 256), (unsigned)C), dim3(256), 0, s, N, (const uint32_t *)boxes_sorted, (const int32_t *)visible_dev, cum_sorted);
    int rc = gc_raster_scan_tiles_views(N, C, cum_sorted, nullptr, cum_sorted, count_dev, w + p.off_scan, p.scan_bytes, (int64_t)p.total, (void *)s);
    if (rc != GC_OK) return rc;
    return gc::check_launch("gc_raster_order_boxes_views");
}

/* Phase 2 on the output of gc_raster_order_boxes_views (always sync-free): emission in depth order from the SORTED boxes (no gather), the tile
 * radix passes, tile_bins.  workspace >= gc_raster_bin_views_workspace_bytes(M_cap, C). */
int gc_raster_bin_sorted_views(int64_t N, int C, int64_t M_cap, const int32_t *count_dev, int32_t *overflow_dev, const int32_t *visible_dev,
                               const int32_t *depth_order, const uint32_t *boxes_sorted, const int32_t *cum_sorted, int tiles_x, int tiles_y,
                               int32_t *gaussian_ids_sorted, int32_t *tile_bins, void *workspace, size_t workspace_bytes, void *stream)
{
    GC_REQUIRE(N >= 0 && C >= 1 && C <= 65535 && M_cap >= 0 && tile_bins && boxes_sorted && count_dev && overflow_dev && visible_dev, "bad arguments");
    GC_REQUIRE(tiles_x <= 255 && tiles_y <= 255, "packed boxes hold at most 255 x 255 tiles");
    return bin_tiles_impl(N, C, M_cap, count_dev, overflow_dev, depth_order, cum_sorted, nullptr, nullptr, nullptr, boxes_sorted, tiles_x, tiles_y,
                          gaussian_ids_sorted, tile_bins, nullptr, workspace, workspace_bytes, stream, "gc_raster_bin_sorted_views", visible_dev);
}

}  // extern "C"
