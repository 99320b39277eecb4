// hbm_calib.hip -- known-byte-count streaming kernels used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports wide coalesced reads by 2x; other widths "calibrate in your
// own access pattern").  Each kernel copies n elements of W bytes per lane, grid-stride, so reads = writes = n * W bytes.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/ubench/libhbm_calib.so scripts/ubench/hbm_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
template <class V> __global__ __launch_bounds__(256) void k_calib_copy(const V *__restrict__ a, V *__restrict__ b, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) b[i] = a[i];
}
struct f3 { float x, y, z; };
}  // namespace

// width: 4, 8, 12 or 16 bytes per lane; bytes: total bytes to copy (multiple of width)
extern "C" int hbm_calib_copy(const void *a, void *b, int64_t bytes, int width, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = bytes / width;
    const unsigned grid = (unsigned)((n + 255) / 256 < 256 * 64 ? (n + 255) / 256 : 256 * 64);
    if (width == 16) hipLaunchKernelGGL((k_calib_copy<uint4>), dim3(grid), dim3(256), 0, s, (const uint4 *)a, (uint4 *)b, n);
    else if (width == 12) hipLaunchKernelGGL((k_calib_copy<f3>), dim3(grid), dim3(256), 0, s, (const f3 *)a, (f3 *)b, n);
    else if (width == 8) hipLaunchKernelGGL((k_calib_copy<uint2>), dim3(grid), dim3(256), 0, s, (const uint2 *)a, (uint2 *)b, n);
    else if (width == 4) hipLaunchKernelGGL((k_calib_copy<float>), dim3(grid), dim3(256), 0, s, (const float *)a, (float *)b, n);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
