// raster_bin.hip -- tile binning for the splat rasterizer on gfx950.
//
// Replaces the inside of gsplat 0.1.3's rasterize_gaussians up to the compositing kernel
// (cumsum -> map_gaussian_to_intersects -> sort -> get_tile_bin_edges), reached from
// /root/reference/gaussctrl/gc_model.py:174-186,191-202.  SURVEY.md Appendix A.3.
//
//   scan      : hand-written 3-launch wave64 scan (2048 elements / workgroup)
//   map       : 1 lane per Gaussian writes its (tile<<32 | depth_bits, id) pairs
//   sort      : stable LSD radix sort of the 64-bit keys restricted to the significant bits
//               (32 depth bits + ceil(log2(tiles+1)) tile bits); rocPRIM device primitive for
//               round 1 (library sort = plumbing here; the hand-written depth-major/tile-minor
//               two-level binning that replaces it is described in DESIGN.md "next").
//   tile_bins : 1 lane per intersection, boundary detection
#include "common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int TILE = 16;
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;   // 2048

__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// block-wide inclusive scan of one int per thread (256 threads = 4 waves); returns inclusive value,
// *total = block sum.
__device__ __forceinline__ int block_incl_scan(int v, int *total)
{
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int s = wave_incl_scan(v);
    if (lane == 63) wsum[wid] = s;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wid) off += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return s + off;
}

// The three scan kernels take the view (camera) index from blockIdx.y (round 5: batched views): in / order / out are [C][N], the block
// sums of view c live `ws` words after those of view c - 1, count[c].  order != NULL: the input is gathered through it (in[order[i]]) --
// the tile counts in depth order without a materialised gather pass.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_local(int64_t N, const int32_t *__restrict__ in, const int32_t *__restrict__ order,
                                                             int32_t *__restrict__ out, int32_t *__restrict__ block_sums, int64_t ws)
{
    in += blockIdx.y * N; out += blockIdx.y * N; block_sums += blockIdx.y * ws;
    if (order) order += blockIdx.y * N;
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int run = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + k;
        int x = i < N ? (order ? in[order[i]] : in[i]) : 0;
        run += x;
        v[k] = run;
    }
    int total;
    int incl = block_incl_scan(run, &total);
    int excl = incl - run;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + k;
        if (i < N) out[i] = v[k] + excl;
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single workgroup per view: exclusive scan of the block sums in place; writes the grand total.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(int nblocks, int32_t *__restrict__ block_sums,
                                                            int32_t *__restrict__ count, int64_t ws)
{
    block_sums += blockIdx.y * ws; count += blockIdx.y;
    int carry = 0;
    for (int base = 0; base < nblocks; base += SCAN_THREADS) {
        int i = base + threadIdx.x;
        int x = i < nblocks ? block_sums[i] : 0;
        int total;
        int incl = block_incl_scan(x, &total);
        if (i < nblocks) block_sums[i] = carry + incl - x;
        carry += total;
    }
    if (threadIdx.x == 0) *count = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_add(int64_t N, int32_t *__restrict__ out,
                                                           const int32_t *__restrict__ block_sums, int64_t ws)
{
    out += blockIdx.y * N; block_sums += blockIdx.y * ws;
    const int add = block_sums[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + (int64_t)k * SCAN_THREADS + threadIdx.x;
        if (i < N) out[i] += add;
    }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void k_map_intersects(int64_t N, int64_t M_cap, const float *__restrict__ xys,
                                                        const float *__restrict__ depths, const int32_t *__restrict__ radii,
                                                        const int32_t *__restrict__ cum, int tiles_x, int tiles_y,
                                                        int64_t *__restrict__ keys, int32_t *__restrict__ ids)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int r = radii[i];
    if (r <= 0) return;
    // same float expressions as the projection kernel / oracle (bit-exact tile box)
    float tcx = xys[2 * i] / (float)TILE, tcy = xys[2 * i + 1] / (float)TILE, tr = (float)r / (float)TILE;
    int minx = clampi((int)(tcx - tr), 0, tiles_x), maxx = clampi((int)(tcx + tr + 1.f), 0, tiles_x);
    int miny = clampi((int)(tcy - tr), 0, tiles_y), maxy = clampi((int)(tcy + tr + 1.f), 0, tiles_y);
    int64_t cur = (i == 0) ? 0 : cum[i - 1];
    const uint32_t dbits = __float_as_uint(depths[i]);
    for (int ty = miny; ty < maxy; ++ty)
        for (int tx = minx; tx < maxx; ++tx) {
            if (cur < M_cap) {
                int64_t tile = (int64_t)ty * tiles_x + tx;
                keys[cur] = (tile << 32) | (int64_t)dbits;
                ids[cur] = (int32_t)i;
            }
            ++cur;
        }
}

// pad [M, M_cap) with a key that sorts after every real key
__global__ __launch_bounds__(256) void k_pad_intersects(int64_t M_cap, const int32_t *__restrict__ count, int num_tiles,
                                                        int64_t *__restrict__ keys, int32_t *__restrict__ ids)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M_cap || i < (int64_t)*count) return;
    keys[i] = (int64_t)num_tiles << 32;
    ids[i] = 0;
}

__global__ __launch_bounds__(256) void k_tile_bins(int64_t M, int num_tiles, const int64_t *__restrict__ keys,
                                                   int32_t *__restrict__ bins)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    int t = (int)(keys[i] >> 32);
    if (t >= num_tiles) return;   // padding
    if (i == 0) bins[2 * t] = 0;
    else {
        int tp = (int)(keys[i - 1] >> 32);
        if (tp != t) { bins[2 * tp + 1] = (int32_t)i; bins[2 * t] = (int32_t)i; }
    }
    if (i == M - 1) bins[2 * t + 1] = (int32_t)M;
    else {
        int tn = (int)(keys[i + 1] >> 32);
        if (tn >= num_tiles) bins[2 * t + 1] = (int32_t)(i + 1);   // next is padding
    }
}

int key_bits(int num_tiles)
{
    int b = 0;
    while ((1 << b) <= num_tiles) ++b;   // tile ids 0..num_tiles (num_tiles = padding key)
    return 32 + b;
}

}  // namespace

extern "C" {

size_t gc_raster_scan_workspace_bytes(int64_t N) { return sizeof(int32_t) * (size_t)(gc::cdiv(N > 0 ? N : 1, SCAN_CHUNK) + 1); }

int gc_raster_scan_tiles_views(int64_t N, int C, const int32_t *num_tiles_hit, const int32_t *order, int32_t *cum_tiles_hit, int32_t *count_dev,
                               void *workspace, size_t workspace_bytes, int64_t ws_view_stride, void *stream)
{
    GC_REQUIRE(N >= 0 && C >= 1 && count_dev && ws_view_stride % 4 == 0, "bad arguments");
    if (N == 0) return hipMemsetAsync(count_dev, 0, 4 * (size_t)C, gc::S(stream)) == hipSuccess ? GC_OK : GC_ELAUNCH;
    if (workspace_bytes < gc_raster_scan_workspace_bytes(N)) { gc::set_error("gc_raster_scan_tiles_views: workspace too small"); return GC_ENOSPC; }
    GC_REQUIRE(C == 1 || (size_t)ws_view_stride >= gc_raster_scan_workspace_bytes(N), "per-view scan scratch regions overlap");
    int nb = (int)gc::cdiv(N, SCAN_CHUNK);
    int32_t *sums = (int32_t *)workspace;
    const int64_t ws = ws_view_stride / 4;
    hipLaunchKernelGGL(k_scan_local, dim3(nb, C), dim3(SCAN_THREADS), 0, gc::S(stream), N, num_tiles_hit, order, cum_tiles_hit, sums, ws);
    hipLaunchKernelGGL(k_scan_sums, dim3(1, C), dim3(SCAN_THREADS), 0, gc::S(stream), nb, sums, count_dev, ws);
    hipLaunchKernelGGL(k_scan_add, dim3(nb, C), dim3(SCAN_THREADS), 0, gc::S(stream), N, cum_tiles_hit, sums, ws);
    return gc::check_launch("gc_raster_scan_tiles_views");
}

int gc_raster_scan_tiles(int64_t N, const int32_t *num_tiles_hit, int32_t *cum_tiles_hit, int32_t *count_dev,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    GC_REQUIRE(N >= 0 && count_dev, "bad arguments");
    return gc_raster_scan_tiles_views(N, 1, num_tiles_hit, nullptr, cum_tiles_hit, count_dev, workspace, workspace_bytes, 0, stream);
}

int gc_raster_read_count(const int32_t *count_dev, int32_t *count_host, void *stream)
{
    if (hipMemcpyAsync(count_host, count_dev, 4, hipMemcpyDeviceToHost, gc::S(stream)) != hipSuccess ||
        hipStreamSynchronize(gc::S(stream)) != hipSuccess) {
        gc::set_error("gc_raster_read_count: %s", hipGetErrorString(hipGetLastError()));
        return GC_ELAUNCH;
    }
    return GC_OK;
}

int gc_raster_map_intersects(int64_t N, int64_t M_cap, const float *xys, const float *depths, const int32_t *radii,
                             const int32_t *cum_tiles_hit, int tiles_x, int tiles_y, int64_t *isect_ids,
                             int32_t *gaussian_ids, void *stream)
{
    if (N == 0) return GC_OK;
    hipLaunchKernelGGL(k_map_intersects, dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), N, M_cap, xys, depths,
                       radii, cum_tiles_hit, tiles_x, tiles_y, isect_ids, gaussian_ids);
    return gc::check_launch("gc_raster_map_intersects");
}

int gc_raster_pad_intersects(int64_t M_cap, const int32_t *count_dev, int num_tiles, int64_t *isect_ids,
                             int32_t *gaussian_ids, void *stream)
{
    if (M_cap == 0) return GC_OK;
    hipLaunchKernelGGL(k_pad_intersects, dim3(gc::cdiv(M_cap, 256)), dim3(256), 0, gc::S(stream), M_cap, count_dev,
                       num_tiles, isect_ids, gaussian_ids);
    return gc::check_launch("gc_raster_pad_intersects");
}

size_t gc_raster_sort_workspace_bytes(int64_t M, int num_tiles)
{
    if (M <= 0) return 0;
    size_t bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)M, 0u,
                                             (unsigned)key_bits(num_tiles), (hipStream_t)0);
    if (e != hipSuccess) { gc::set_error("gc_raster_sort_workspace_bytes: %s", hipGetErrorString(e)); return 0; }
    return bytes + 256;
}

int gc_raster_sort_intersects(int64_t M, int num_tiles, const int64_t *isect_ids, const int32_t *gaussian_ids,
                              int64_t *isect_ids_sorted, int32_t *gaussian_ids_sorted, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    if (M <= 0) return GC_OK;
    size_t bytes = workspace_bytes;
    hipError_t e = rocprim::radix_sort_pairs(workspace, bytes, (const uint64_t *)isect_ids, (uint64_t *)isect_ids_sorted,
                                             gaussian_ids, gaussian_ids_sorted, (size_t)M, 0u,
                                             (unsigned)key_bits(num_tiles), gc::S(stream));
    if (e != hipSuccess) { gc::set_error("gc_raster_sort_intersects: %s", hipGetErrorString(e)); return GC_ELAUNCH; }
    return GC_OK;
}

int gc_raster_tile_bins(int64_t M, int num_tiles, const int64_t *isect_ids_sorted, int32_t *tile_bins, void *stream)
{
    if (hipMemsetAsync(tile_bins, 0, sizeof(int32_t) * 2 * (size_t)num_tiles, gc::S(stream)) != hipSuccess) return GC_ELAUNCH;
    if (M <= 0) return GC_OK;
    hipLaunchKernelGGL(k_tile_bins, dim3(gc::cdiv(M, 256)), dim3(256), 0, gc::S(stream), M, num_tiles, isect_ids_sorted,
                       tile_bins);
    return gc::check_launch("gc_raster_tile_bins");
}

}  // extern "C"
