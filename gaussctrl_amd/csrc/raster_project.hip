// raster_project.hip -- per-Gaussian front end of the splat rasterizer for gfx950.
//
// Replaces gsplat 0.1.3's project_gaussians / spherical_harmonics (+ their backward) as called from
// /root/reference/gaussctrl/gc_model.py:140-154,166 and adds the fused "one pass over the 59-float
// parameter record" kernels used by the product path (gc_model.py:138-169,181 in one launch).
//
// HBM-bound, 1 lane per Gaussian, wave64, no LDS needed: the record is read once with 16-byte loads
// where alignment allows.  This translation unit is compiled with -ffp-contract=off: the projection
// arithmetic is written as explicit IEEE binary32 operations in a fixed order so that the integer
// outputs (radii, tile boxes, num_tiles_hit -> sort keys) are bit-identical to the CPU oracle.
#include "common.h"

namespace {

constexpr int TILE = 16;

struct Cam {
    float V[12];   // row-major 3x4 world->camera
    float P[16];   // row-major 4x4 full projection
    float fx, fy, cx, cy;
    int H, W, tiles_x, tiles_y;
    float clip, glob;
    float ox, oy, oz;   // camera origin (world) for view directions
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Rot {
    float R[9];
    float qn[4];
    float inv_norm;
};

__device__ __forceinline__ void quat_to_rotmat(float w, float x, float y, float z, Rot &o)
{
    float n2 = ((w * w + x * x) + y * y) + z * z;
    float s = 1.f / sqrtf(n2);
    w = w * s; x = x * s; y = y * s; z = z * s;
    o.qn[0] = w; o.qn[1] = x; o.qn[2] = y; o.qn[3] = z; o.inv_norm = s;
    o.R[0] = 1.f - 2.f * (y * y + z * z);
    o.R[1] = 2.f * (x * y - w * z);
    o.R[2] = 2.f * (x * z + w * y);
    o.R[3] = 2.f * (x * y + w * z);
    o.R[4] = 1.f - 2.f * (x * x + z * z);
    o.R[5] = 2.f * (y * z - w * x);
    o.R[6] = 2.f * (x * z - w * y);
    o.R[7] = 2.f * (y * z + w * x);
    o.R[8] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void scale_rot_to_cov3d(float s0, float s1, float s2, float glob, const float *R,
                                                   float *c, float *m)
{
    float sx = glob * s0, sy = glob * s1, sz = glob * s2;
    m[0] = R[0] * sx; m[1] = R[1] * sy; m[2] = R[2] * sz;
    m[3] = R[3] * sx; m[4] = R[4] * sy; m[5] = R[5] * sz;
    m[6] = R[6] * sx; m[7] = R[7] * sy; m[8] = R[8] * sz;
    c[0] = (m[0] * m[0] + m[1] * m[1]) + m[2] * m[2];
    c[1] = (m[0] * m[3] + m[1] * m[4]) + m[2] * m[5];
    c[2] = (m[0] * m[6] + m[1] * m[7]) + m[2] * m[8];
    c[3] = (m[3] * m[3] + m[4] * m[4]) + m[5] * m[5];
    c[4] = (m[3] * m[6] + m[4] * m[7]) + m[5] * m[8];
    c[5] = (m[6] * m[6] + m[7] * m[7]) + m[8] * m[8];
}

struct Proj {
    float cov3d[6];
    float xy[2];
    float depth;
    float conic[3];
    int radius;
    int tiles_hit;
};

// SURVEY.md Appendix A.1; op order mirrors oracle/raster_ref.c::project_one exactly.
__device__ __forceinline__ bool project_one(const Cam &cam, float p0, float p1, float p2, float s0, float s1,
                                            float s2, float qw, float qx, float qy, float qz, Proj &o)
{
    const float *V = cam.V, *P = cam.P;
    o.radius = 0; o.tiles_hit = 0; o.depth = 0.f; o.xy[0] = o.xy[1] = 0.f;
    o.conic[0] = o.conic[1] = o.conic[2] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) o.cov3d[k] = 0.f;
    float tx = ((V[0] * p0 + V[1] * p1) + V[2] * p2) + V[3];
    float ty = ((V[4] * p0 + V[5] * p1) + V[6] * p2) + V[7];
    float tz = ((V[8] * p0 + V[9] * p1) + V[10] * p2) + V[11];
    if (tz <= cam.clip) return false;
    Rot rot; float m[9];
    quat_to_rotmat(qw, qx, qy, qz, rot);
    scale_rot_to_cov3d(s0, s1, s2, cam.glob, rot.R, o.cov3d, m);
    float lim_x = 1.3f * (0.5f * (float)cam.W / cam.fx);
    float lim_y = 1.3f * (0.5f * (float)cam.H / cam.fy);
    float txc = tz * clampf(tx / tz, -lim_x, lim_x);
    float tyc = tz * clampf(ty / tz, -lim_y, lim_y);
    float rz = 1.f / tz;
    float rz2 = rz * rz;
    float j00 = cam.fx * rz, j02 = -(cam.fx * txc) * rz2;
    float j11 = cam.fy * rz, j12 = -(cam.fy * tyc) * rz2;
    float t00 = j00 * V[0] + j02 * V[8], t01 = j00 * V[1] + j02 * V[9], t02 = j00 * V[2] + j02 * V[10];
    float t10 = j11 * V[4] + j12 * V[8], t11 = j11 * V[5] + j12 * V[9], t12 = j11 * V[6] + j12 * V[10];
    const float *c = o.cov3d;
    float u00 = (t00 * c[0] + t01 * c[1]) + t02 * c[2];
    float u01 = (t00 * c[1] + t01 * c[3]) + t02 * c[4];
    float u02 = (t00 * c[2] + t01 * c[4]) + t02 * c[5];
    float u10 = (t10 * c[0] + t11 * c[1]) + t12 * c[2];
    float u11 = (t10 * c[1] + t11 * c[3]) + t12 * c[4];
    float u12 = (t10 * c[2] + t11 * c[4]) + t12 * c[5];
    float a = ((u00 * t00 + u01 * t01) + u02 * t02) + 0.3f;
    float b = (u00 * t10 + u01 * t11) + u02 * t12;
    float d = ((u10 * t10 + u11 * t11) + u12 * t12) + 0.3f;
    float det = a * d - b * b;
    if (det == 0.f) return false;
    float inv_det = 1.f / det;
    // written before the tile-box cull, like gsplat / the oracle (unobservable for culled splats)
    o.conic[0] = d * inv_det; o.conic[1] = -b * inv_det; o.conic[2] = a * inv_det;
    float mid = 0.5f * (a + d);
    float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    float v1 = mid + disc, v2 = mid - disc;
    float radius = ceilf(3.f * sqrtf(fmaxf(v1, v2)));
    float hx = ((P[0] * p0 + P[1] * p1) + P[2] * p2) + P[3];
    float hy = ((P[4] * p0 + P[5] * p1) + P[6] * p2) + P[7];
    float hw = ((P[12] * p0 + P[13] * p1) + P[14] * p2) + P[15];
    float rw = 1.f / (hw + 1e-6f);
    float px = (0.5f * (float)cam.W) * (hx * rw) + cam.cx - 0.5f;
    float py = (0.5f * (float)cam.H) * (hy * rw) + cam.cy - 0.5f;
    float tcx = px / (float)TILE, tcy = py / (float)TILE, tr = radius / (float)TILE;
    int minx = clampi((int)(tcx - tr), 0, cam.tiles_x), maxx = clampi((int)(tcx + tr + 1.f), 0, cam.tiles_x);
    int miny = clampi((int)(tcy - tr), 0, cam.tiles_y), maxy = clampi((int)(tcy + tr + 1.f), 0, cam.tiles_y);
    int area = (maxx - minx) * (maxy - miny);
    if (area <= 0) return false;
    o.tiles_hit = area; o.depth = tz; o.radius = (int)radius;
    o.xy[0] = px; o.xy[1] = py;
    return true;
}

struct ProjGrad {
    float vm[3], vs[3], vq[4];
};

// True VJP of project_one w.r.t. (mean, scale, raw quat); mirrors oracle orc_project_gaussians_bwd.
__device__ __forceinline__ void project_one_bwd(const Cam &cam, float p0, float p1, float p2, float s0, float s1,
                                                float s2, float qw, float qx, float qy, float qz,
                                                float X00, float X01, float X11, float vx, float vy, float vz,
                                                float g0, float g1, float g2, ProjGrad &o)
{
    const float *V = cam.V, *P = cam.P;
    float hx = ((P[0] * p0 + P[1] * p1) + P[2] * p2) + P[3];
    float hy = ((P[4] * p0 + P[5] * p1) + P[6] * p2) + P[7];
    float hw = ((P[12] * p0 + P[13] * p1) + P[14] * p2) + P[15];
    float rw = 1.f / (hw + 1e-6f);
    float vnx = 0.5f * (float)cam.W * vx, vny = 0.5f * (float)cam.H * vy;
    float vhx = vnx * rw, vhy = vny * rw, vhw = -(vnx * hx + vny * hy) * rw * rw;
    o.vm[0] = P[0] * vhx + P[4] * vhy + P[12] * vhw;
    o.vm[1] = P[1] * vhx + P[5] * vhy + P[13] * vhw;
    o.vm[2] = P[2] * vhx + P[6] * vhy + P[14] * vhw;
    o.vm[0] += V[8] * vz; o.vm[1] += V[9] * vz; o.vm[2] += V[10] * vz;
    float G00 = g0, G01 = 0.5f * g1, G11 = g2;
    float a00 = X00 * G00 + X01 * G01, a01 = X00 * G01 + X01 * G11;
    float a10 = X01 * G00 + X11 * G01, a11 = X01 * G01 + X11 * G11;
    float C00 = -(a00 * X00 + a01 * X01), C01 = -(a00 * X01 + a01 * X11), C11 = -(a10 * X01 + a11 * X11);
    float tx = ((V[0] * p0 + V[1] * p1) + V[2] * p2) + V[3];
    float ty = ((V[4] * p0 + V[5] * p1) + V[6] * p2) + V[7];
    float tz = ((V[8] * p0 + V[9] * p1) + V[10] * p2) + V[11];
    float lim_x = 1.3f * (0.5f * (float)cam.W / cam.fx), lim_y = 1.3f * (0.5f * (float)cam.H / cam.fy);
    float rx = tx / tz, ry = ty / tz;
    float rxc = clampf(rx, -lim_x, lim_x), ryc = clampf(ry, -lim_y, lim_y);
    bool clx = (rx != rxc), cly = (ry != ryc);
    float txc = tz * rxc, tyc = tz * ryc;
    float rz = 1.f / tz, rz2 = rz * rz;
    float fx = cam.fx, fy = cam.fy;
    float j00 = fx * rz, j02 = -(fx * txc) * rz2, j11 = fy * rz, j12 = -(fy * tyc) * rz2;
    float T[6] = {j00 * V[0] + j02 * V[8], j00 * V[1] + j02 * V[9], j00 * V[2] + j02 * V[10],
                  j11 * V[4] + j12 * V[8], j11 * V[5] + j12 * V[9], j11 * V[6] + j12 * V[10]};
    Rot rot; float M[9], c3[6];
    quat_to_rotmat(qw, qx, qy, qz, rot);
    scale_rot_to_cov3d(s0, s1, s2, cam.glob, rot.R, c3, M);
    float S[9] = {c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]};
    float GT[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { GT[k] = C00 * T[k] + C01 * T[3 + k]; GT[3 + k] = C01 * T[k] + C11 * T[3 + k]; }
    float vS[9], vT[6], vM[9], vR[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) vS[3 * r + k] = T[r] * GT[k] + T[3 + r] * GT[3 + k];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            vT[3 * r + k] = 2.f * (GT[3 * r] * S[k] + GT[3 * r + 1] * S[3 + k] + GT[3 * r + 2] * S[6 + k]);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            vM[3 * r + k] = 2.f * (vS[3 * r] * M[k] + vS[3 * r + 1] * M[3 + k] + vS[3 * r + 2] * M[6 + k]);
    float sc[3] = {s0, s1, s2};
    const float *R = rot.R;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) vR[3 * r + k] = vM[3 * r + k] * (cam.glob * sc[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) o.vs[k] = cam.glob * (R[k] * vM[k] + R[3 + k] * vM[3 + k] + R[6 + k] * vM[6 + k]);
    float w = rot.qn[0], x = rot.qn[1], y = rot.qn[2], z = rot.qn[3];
    float vqn[4];
    vqn[0] = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
    vqn[1] = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
    vqn[2] = 2.f * (x * (vR[1] + vR[3]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
    vqn[3] = 2.f * (x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
    float dotq = rot.qn[0] * vqn[0] + rot.qn[1] * vqn[1] + rot.qn[2] * vqn[2] + rot.qn[3] * vqn[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) o.vq[k] = (vqn[k] - rot.qn[k] * dotq) * rot.inv_norm;
    float vj00 = vT[0] * V[0] + vT[1] * V[1] + vT[2] * V[2];
    float vj02 = vT[0] * V[8] + vT[1] * V[9] + vT[2] * V[10];
    float vj11 = vT[3] * V[4] + vT[4] * V[5] + vT[5] * V[6];
    float vj12 = vT[3] * V[8] + vT[4] * V[9] + vT[5] * V[10];
    float v_rz = fx * vj00 + fy * vj11 - 2.f * rz * (fx * txc * vj02 + fy * tyc * vj12);
    float v_txc = -fx * rz2 * vj02, v_tyc = -fy * rz2 * vj12;
    float v_tz = -rz2 * v_rz, v_tx = 0.f, v_ty = 0.f;
    if (clx) v_tz += rxc * v_txc; else v_tx += v_txc;
    if (cly) v_tz += ryc * v_tyc; else v_ty += v_tyc;
    o.vm[0] += V[0] * v_tx + V[4] * v_ty + V[8] * v_tz;
    o.vm[1] += V[1] * v_tx + V[5] * v_ty + V[9] * v_tz;
    o.vm[2] += V[2] * v_tx + V[6] * v_ty + V[10] * v_tz;
}

// ---------------------------------------------------------------- SH (Appendix A.2)
__device__ __forceinline__ void sh_basis(int n, float x, float y, float z, float *B)
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
#pragma unroll
    for (int k = 0; k < 16; ++k) B[k] = 0.f;
    B[0] = C0;
    if (n < 1) return;
    B[1] = -C1 * y; B[2] = C1 * z; B[3] = -C1 * x;
    if (n < 2) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = 1.0925484305920792f * xy; B[5] = -1.0925484305920792f * yz;
    B[6] = 0.31539156525252005f * (2.f * zz - xx - yy);
    B[7] = -1.0925484305920792f * xz; B[8] = 0.5462742152960396f * (xx - yy);
    if (n < 3) return;
    B[9] = -0.5900435899266435f * y * (3.f * xx - yy);
    B[10] = 2.890611442640554f * xy * z;
    B[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
    B[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
    B[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
    B[14] = 1.445305721320277f * z * (xx - yy);
    B[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
}

// ---------------------------------------------------------------- kernels: gsplat operator surface
__global__ __launch_bounds__(256) void k_project_fwd(int64_t N, Cam cam, const float *__restrict__ means,
                                                     const float *__restrict__ scales, const float *__restrict__ quats,
                                                     float *__restrict__ cov3d, float *__restrict__ xys,
                                                     float *__restrict__ depths, int32_t *__restrict__ radii,
                                                     float *__restrict__ conics, int32_t *__restrict__ tiles_hit)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Proj o;
    project_one(cam, means[3 * i], means[3 * i + 1], means[3 * i + 2], scales[3 * i], scales[3 * i + 1],
                scales[3 * i + 2], quats[4 * i], quats[4 * i + 1], quats[4 * i + 2], quats[4 * i + 3], o);
#pragma unroll
    for (int k = 0; k < 6; ++k) cov3d[6 * i + k] = o.cov3d[k];
    xys[2 * i] = o.xy[0]; xys[2 * i + 1] = o.xy[1];
    depths[i] = o.depth; radii[i] = o.radius; tiles_hit[i] = o.tiles_hit;
    conics[3 * i] = o.conic[0]; conics[3 * i + 1] = o.conic[1]; conics[3 * i + 2] = o.conic[2];
}

__global__ __launch_bounds__(256) void k_project_bwd(int64_t N, Cam cam, const float *__restrict__ means,
                                                     const float *__restrict__ scales, const float *__restrict__ quats,
                                                     const int32_t *__restrict__ radii, const float *__restrict__ conics,
                                                     const float *__restrict__ v_xy, const float *__restrict__ v_depth,
                                                     const float *__restrict__ v_conic, float *__restrict__ v_mean,
                                                     float *__restrict__ v_scale, float *__restrict__ v_quat)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    ProjGrad g;
#pragma unroll
    for (int k = 0; k < 3; ++k) { g.vm[k] = 0.f; g.vs[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) g.vq[k] = 0.f;
    if (radii[i] > 0)
        project_one_bwd(cam, means[3 * i], means[3 * i + 1], means[3 * i + 2], scales[3 * i], scales[3 * i + 1],
                        scales[3 * i + 2], quats[4 * i], quats[4 * i + 1], quats[4 * i + 2], quats[4 * i + 3],
                        conics[3 * i], conics[3 * i + 1], conics[3 * i + 2], v_xy[2 * i], v_xy[2 * i + 1],
                        v_depth ? v_depth[i] : 0.f, v_conic[3 * i], v_conic[3 * i + 1], v_conic[3 * i + 2], g);
#pragma unroll
    for (int k = 0; k < 3; ++k) { v_mean[3 * i + k] = g.vm[k]; v_scale[3 * i + k] = g.vs[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) v_quat[4 * i + k] = g.vq[k];
}

__global__ __launch_bounds__(256) void k_sh_fwd(int64_t N, int K, int n, const float *__restrict__ dirs,
                                                const float *__restrict__ coeffs, float *__restrict__ colors)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float B[16];
    sh_basis(n, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], B);
    int Ku = (n + 1) * (n + 1);
    float acc[3] = {0.f, 0.f, 0.f};
    const float *c = coeffs + (size_t)i * K * 3;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < Ku) {
            acc[0] += B[k] * c[3 * k]; acc[1] += B[k] * c[3 * k + 1]; acc[2] += B[k] * c[3 * k + 2];
        }
    colors[3 * i] = acc[0]; colors[3 * i + 1] = acc[1]; colors[3 * i + 2] = acc[2];
}

__global__ __launch_bounds__(256) void k_sh_bwd(int64_t N, int K, int n, const float *__restrict__ dirs,
                                                const float *__restrict__ v_colors, float *__restrict__ v_coeffs)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float B[16];
    sh_basis(n, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], B);
    int Ku = (n + 1) * (n + 1);
    float v0 = v_colors[3 * i], v1 = v_colors[3 * i + 1], v2 = v_colors[3 * i + 2];
    float *o = v_coeffs + (size_t)i * K * 3;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < K) {
            float b = k < Ku ? B[k] : 0.f;
            o[3 * k] = b * v0; o[3 * k + 1] = b * v1; o[3 * k + 2] = b * v2;
        }
}

// ---------------------------------------------------------------- kernels: fused product path
// One pass over the 59-float record (236 B/Gaussian read, 48 B written).  Mirrors the reference
// op-by-op: exp(scales), q/|q| (gc_model.py:144), project, viewdirs (gc_model.py:163-164),
// SH, clamp(+0.5,min 0) (:167) -- or sigmoid(features_dc) when config.sh_degree == 0 (:169, n_use = -1) -- and sigmoid(opacity) (:181).
//
// HBM access: one lane per Gaussian, but the 45-float features_rest record (180 of the 236 bytes) is NOT read lane-by-lane --
// a 180-byte lane stride makes every load instruction touch 64 different cache lines (rocprofv3 FETCH_SIZE of the round-1 kernel:
// 3.6x the algorithmic bytes).  The workgroup's 256 records are one contiguous 46 KB block: it is streamed with 16-byte-per-lane
// Tight tile box of the fused path (round 3).  gsplat bins a Gaussian into every tile of the box around a CIRCLE of 3 sqrt(lambda_max);
// a pixel can only pass the compositing test alpha = opacity * exp(-sigma) >= 1/255 inside the ellipse sigma <= tau = ln(255 opacity),
// whose axis-aligned bounding box has half extents sqrt(2 tau cyy / det), sqrt(2 tau cxx / det) (conic = inverse covariance).  For
// anisotropic or faint Gaussians that box is much smaller: intersected with gsplat's box it drops 31 % of the (tile, Gaussian) pairs
// of the synthetic scenes (the exact per-tile test: 37 %) at O(1) per Gaussian.  tau carries the margin of raster_composite.hip's block
// test and the extents a relative + absolute slack, far above the rounding of either side; every tile that is dropped contains no pixel
// that passes the per-pixel test, so images and gradients are bit-identical to the gsplat box.  Packed min_x | max_x << 8 | min_y << 16 |
// max_y << 24 (exclusive maxima; tiles_x, tiles_y <= 255); 0 = no tile.
__device__ __forceinline__ uint32_t tight_tile_box(const Cam &cam, const Proj &o, float opacity)
{
    const float px = o.xy[0], py = o.xy[1];
    const float tcx = px / (float)TILE, tcy = py / (float)TILE, tr = (float)o.radius / (float)TILE;      // gsplat's box (same expressions as project_one)
    int minx = clampi((int)(tcx - tr), 0, cam.tiles_x), maxx = clampi((int)(tcx + tr + 1.f), 0, cam.tiles_x);
    int miny = clampi((int)(tcy - tr), 0, cam.tiles_y), maxy = clampi((int)(tcy + tr + 1.f), 0, cam.tiles_y);
    const float cxx = o.conic[0], cxy = o.conic[1], cyy = o.conic[2];
    const float det = cxx * cyy - cxy * cxy;
    const float tau = logf(255.f * opacity) * 1.001f + 0.01f;
    if (tau < 0.f) return 0u;                                            // can never reach 1/255 anywhere
    if (cxx > 0.f && cyy > 0.f && det > 0.f && tau >= 0.f) {             // (NaN / degenerate conic: keep gsplat's box)
        const float ex = sqrtf(2.f * tau * cyy / det) * 1.0001f + 1e-3f, ey = sqrtf(2.f * tau * cxx / det) * 1.0001f + 1e-3f;
        const float x0 = ceilf(px - ex), x1 = floorf(px + ex), y0 = ceilf(py - ey), y1 = floorf(py + ey);      // pixel centres inside
        if (!(x0 <= x1 && y0 <= y1)) return 0u;
        const int tx0 = (int)floorf(x0 / (float)TILE), tx1 = (int)floorf(x1 / (float)TILE) + 1;
        const int ty0 = (int)floorf(y0 / (float)TILE), ty1 = (int)floorf(y1 / (float)TILE) + 1;
        minx = minx > tx0 ? minx : tx0; maxx = maxx < tx1 ? maxx : tx1;
        miny = miny > ty0 ? miny : ty0; maxy = maxy < ty1 ? maxy : ty1;
    }
    if (maxx <= minx || maxy <= miny) return 0u;
    return (uint32_t)minx | ((uint32_t)maxx << 8) | ((uint32_t)miny << 16) | ((uint32_t)maxy << 24);
}

// loads into LDS, and each lane then reads ITS record from LDS at a 45-dword stride (odd: bank-conflict free).
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

template <int K>
__global__ __launch_bounds__(256) void k_project_sh_fwd(int64_t N, Cam cam, int n_use,
                                                        const float *__restrict__ means, const float *__restrict__ log_scales,
                                                        const float *__restrict__ quats, const float *__restrict__ op_logit,
                                                        const float *__restrict__ f_dc, const float *__restrict__ f_rest,
                                                        float *__restrict__ xys, float *__restrict__ depths,
                                                        int32_t *__restrict__ radii, float *__restrict__ conics,
                                                        int32_t *__restrict__ tiles_hit, float *__restrict__ rgbs,
                                                        float *__restrict__ opac, uint32_t *__restrict__ tile_box)
{
    constexpr int R = (K - 1) * 3;                         // floats of features_rest per Gaussian
    __shared__ __attribute__((aligned(16))) float srest[R > 0 ? 256 * R : 4];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int64_t i = i0 + tid;
    // the lane's own record is requested BEFORE the cooperative staging of the SH block, so that the two HBM round trips overlap
    // (3 workgroups per CU -- the 46 KB of LDS -- leave little else to hide them behind)
    const int64_t ic = i < N ? i : N - 1;
    float p0 = means[3 * ic], p1 = means[3 * ic + 1], p2 = means[3 * ic + 2];
    const float l0 = log_scales[3 * ic], l1 = log_scales[3 * ic + 1], l2 = log_scales[3 * ic + 2];
    float4 q = *reinterpret_cast<const float4 *>(quats + 4 * ic);
    const float opl = op_logit[ic];
    const float d0 = f_dc[3 * ic], d1 = f_dc[3 * ic + 1], d2 = f_dc[3 * ic + 2];
    if (R > 0 && n_use > 0) {
        const int64_t cnt = ((N - i0 < 256 ? N - i0 : 256)) * R;        // floats of this workgroup's block
        const float *src = f_rest + i0 * R;                              // 256 * R * 4 bytes per block: 16-byte aligned
        for (int64_t j = tid; j < cnt / 4; j += 256) reinterpret_cast<float4 *>(srest)[j] = reinterpret_cast<const float4 *>(src)[j];
        for (int64_t j = (cnt / 4) * 4 + tid; j < cnt; j += 256) srest[j] = src[j];
    }
    Proj o;
    bool ok = false;
    if (i < N) {
        float s0 = expf(l0), s1 = expf(l1), s2 = expf(l2);
        float qn = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
        q.x = q.x / qn; q.y = q.y / qn; q.z = q.z / qn; q.w = q.w / qn;
        ok = project_one(cam, p0, p1, p2, s0, s1, s2, q.x, q.y, q.z, q.w, o);
        xys[2 * i] = o.xy[0]; xys[2 * i + 1] = o.xy[1];
        const float op = sigmoidf(opl);
        if (tile_box) {      // tight tile box (see tight_tile_box): nth and the box the emission walks shrink together
            uint32_t box = 0;
            if (ok) box = tight_tile_box(cam, o, op);
            tile_box[i] = box;
            o.tiles_hit = (int)(((box >> 8) & 255u) - (box & 255u)) * (int)((box >> 24) - ((box >> 16) & 255u));
        }
        depths[i] = o.depth; radii[i] = o.radius; tiles_hit[i] = o.tiles_hit;
        conics[3 * i] = o.conic[0]; conics[3 * i + 1] = o.conic[1]; conics[3 * i + 2] = o.conic[2];
        opac[i] = op;
    }
    if (R > 0 && n_use > 0) __syncthreads();
    if (i >= N) return;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (ok) {
        if (n_use < 0) {          // config.sh_degree == 0: rgbs = sigmoid(features_dc)   (gc_model.py:169)
            c0 = sigmoidf(d0); c1 = sigmoidf(d1); c2 = sigmoidf(d2);
        } else {
            float dx = p0 - cam.ox, dy = p1 - cam.oy, dz = p2 - cam.oz;
            float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
            dx = dx / dn; dy = dy / dn; dz = dz / dn;
            float B[16];
            sh_basis(n_use, dx, dy, dz, B);
            c0 = B[0] * d0; c1 = B[0] * d1; c2 = B[0] * d2;
            const float *r = srest + tid * R;
            int Ku = (n_use + 1) * (n_use + 1);
#pragma unroll
            for (int k = 1; k < K; ++k)
                if (k < Ku) {
                    c0 += B[k] * r[3 * (k - 1)]; c1 += B[k] * r[3 * (k - 1) + 1]; c2 += B[k] * r[3 * (k - 1) + 2];
                }
            c0 = fmaxf(c0 + 0.5f, 0.f); c1 = fmaxf(c1 + 0.5f, 0.f); c2 = fmaxf(c2 + 0.5f, 0.f);
        }
    }
    rgbs[3 * i] = c0; rgbs[3 * i + 1] = c1; rgbs[3 * i + 2] = c2;
}

// Backward of the above.  The colour path needs only the FORWARD colours (rgbs): the clamp(min = 0) mask is rgbs > 0 and the
// sigmoid mode's derivative is s (1 - s), so the 192-byte SH record is not re-read; the 180-byte features_rest gradient is staged in
// LDS and written with 16-byte-per-lane stores (a lane-strided 45-float store has the same 64-lines-per-instruction problem).
// ACC: the six outputs are accumulated into (+=) instead of written -- gradient accumulation over the views of a batch without a
// separate read-add-write pass per tensor (the caller owns zeroing / the first view runs with ACC = false).
template <int K, bool ACC>
__global__ __launch_bounds__(256) void k_project_sh_bwd(int64_t N, Cam cam, int n_use,
                                                        const float *__restrict__ means, const float *__restrict__ log_scales,
                                                        const float *__restrict__ quats, const float *__restrict__ op_logit,
                                                        const float *__restrict__ rgbs,
                                                        const int32_t *__restrict__ radii, const float *__restrict__ conics,
                                                        const float *__restrict__ v_xy, const float *__restrict__ v_conic,
                                                        const float *__restrict__ v_rgbs, const float *__restrict__ v_opac,
                                                        float *__restrict__ v_means, float *__restrict__ v_ls,
                                                        float *__restrict__ v_quats, float *__restrict__ v_oplogit,
                                                        float *__restrict__ v_dc, float *__restrict__ v_rest)
{
    constexpr int R = (K - 1) * 3;
    __shared__ __attribute__((aligned(16))) float svr[R > 0 ? 256 * R : 4];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int64_t i = i0 + tid;
    float *vr = svr + tid * R;
    auto put = [](float *p, float v) __attribute__((always_inline)) { *p = ACC ? *p + v : v; };
    if (i < N) {
        if (radii[i] <= 0) {
            if (!ACC) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { v_means[3 * i + k] = 0.f; v_ls[3 * i + k] = 0.f; v_dc[3 * i + k] = 0.f; }
                *reinterpret_cast<float4 *>(v_quats + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
                v_oplogit[i] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < R; ++k) vr[k] = 0.f;
        } else {
            float p0 = means[3 * i], p1 = means[3 * i + 1], p2 = means[3 * i + 2];
            float s0 = expf(log_scales[3 * i]), s1 = expf(log_scales[3 * i + 1]), s2 = expf(log_scales[3 * i + 2]);
            float4 qr = *reinterpret_cast<const float4 *>(quats + 4 * i);
            float qn = sqrtf(((qr.x * qr.x + qr.y * qr.y) + qr.z * qr.z) + qr.w * qr.w);
            float q0 = qr.x / qn, q1 = qr.y / qn, q2 = qr.z / qn, q3 = qr.w / qn;
            ProjGrad g;
            project_one_bwd(cam, p0, p1, p2, s0, s1, s2, q0, q1, q2, q3, conics[3 * i], conics[3 * i + 1], conics[3 * i + 2],
                            v_xy[2 * i], v_xy[2 * i + 1], 0.f, v_conic[3 * i], v_conic[3 * i + 1], v_conic[3 * i + 2], g);
            put(v_means + 3 * i, g.vm[0]); put(v_means + 3 * i + 1, g.vm[1]); put(v_means + 3 * i + 2, g.vm[2]);
            put(v_ls + 3 * i, g.vs[0] * s0); put(v_ls + 3 * i + 1, g.vs[1] * s1); put(v_ls + 3 * i + 2, g.vs[2] * s2);
            // outer normalisation q/|q| (gc_model.py:144)
            float dq = q0 * g.vq[0] + q1 * g.vq[1] + q2 * g.vq[2] + q3 * g.vq[3];
            float4 vq4 = make_float4((g.vq[0] - q0 * dq) / qn, (g.vq[1] - q1 * dq) / qn, (g.vq[2] - q2 * dq) / qn, (g.vq[3] - q3 * dq) / qn);
            if (ACC) { const float4 o = *reinterpret_cast<const float4 *>(v_quats + 4 * i); vq4.x += o.x; vq4.y += o.y; vq4.z += o.z; vq4.w += o.w; }
            *reinterpret_cast<float4 *>(v_quats + 4 * i) = vq4;
            float op = sigmoidf(op_logit[i]);
            put(v_oplogit + i, v_opac[i] * op * (1.f - op));
            const float r0 = rgbs[3 * i], r1 = rgbs[3 * i + 1], r2 = rgbs[3 * i + 2];
            if (n_use < 0) {        // d sigmoid(features_dc)
                put(v_dc + 3 * i, v_rgbs[3 * i] * r0 * (1.f - r0)); put(v_dc + 3 * i + 1, v_rgbs[3 * i + 1] * r1 * (1.f - r1));
                put(v_dc + 3 * i + 2, v_rgbs[3 * i + 2] * r2 * (1.f - r2));
#pragma unroll
                for (int k = 0; k < R; ++k) vr[k] = 0.f;
            } else {
                float dx = p0 - cam.ox, dy = p1 - cam.oy, dz = p2 - cam.oz;
                float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
                dx = dx / dn; dy = dy / dn; dz = dz / dn;
                float B[16];
                sh_basis(n_use, dx, dy, dz, B);
                int Ku = (n_use + 1) * (n_use + 1);
                // clamp(SH + 0.5, min 0) (gc_model.py:167): the gradient passes where the forward colour is positive
                float v0 = r0 > 0.f ? v_rgbs[3 * i] : 0.f;
                float v1 = r1 > 0.f ? v_rgbs[3 * i + 1] : 0.f;
                float v2 = r2 > 0.f ? v_rgbs[3 * i + 2] : 0.f;
                put(v_dc + 3 * i, B[0] * v0); put(v_dc + 3 * i + 1, B[0] * v1); put(v_dc + 3 * i + 2, B[0] * v2);
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    float b = k < Ku ? B[k] : 0.f;
                    vr[3 * (k - 1)] = b * v0; vr[3 * (k - 1) + 1] = b * v1; vr[3 * (k - 1) + 2] = b * v2;
                }
            }
        }
    }
    if (R > 0) {
        __syncthreads();
        const int64_t cnt = ((N - i0 < 256 ? N - i0 : 256)) * R;
        float *dst = v_rest + i0 * R;
        for (int64_t j = tid; j < cnt / 4; j += 256) {
            float4 v = reinterpret_cast<const float4 *>(svr)[j];
            if (ACC) { const float4 o = reinterpret_cast<const float4 *>(dst)[j]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            reinterpret_cast<float4 *>(dst)[j] = v;
        }
        for (int64_t j = (cnt / 4) * 4 + tid; j < cnt; j += 256) dst[j] = ACC ? dst[j] + svr[j] : svr[j];
    }
}

// ---------------------------------------------------------------- kernels: C views per launch (round 5)
// The 236-byte parameter record of a Gaussian does not depend on the camera: a launch over C views reads it ONCE (the SH block staged
// through LDS once) and projects / shades it for every view of the batch, instead of C launches that each stream the whole record again --
// 236 + C * 60 bytes per Gaussian instead of C * 296.  Same device functions, same operation order as the single-view kernel: every output
// of view c is bit-identical to what gc_project_sh_fwd[_boxes] writes for that camera.  Per-view outputs are [C][N][..]; `opac` (sigmoid of
// the opacity logit, camera independent) is written once, [N]; depth_pairs (optional) are the depth-order sort's input pairs.
constexpr int MAXV = 8;
struct CamBatch { Cam cam[MAXV]; int C; };

template <int K>
__global__ __launch_bounds__(256) void k_project_sh_fwd_views(int64_t N, CamBatch cb, int n_use,
                                                              const float *__restrict__ means, const float *__restrict__ log_scales,
                                                              const float *__restrict__ quats, const float *__restrict__ op_logit,
                                                              const float *__restrict__ f_dc, const float *__restrict__ f_rest,
                                                              float *__restrict__ xys, float *__restrict__ depths,
                                                              int32_t *__restrict__ radii, float *__restrict__ conics,
                                                              int32_t *__restrict__ tiles_hit, float *__restrict__ rgbs,
                                                              float *__restrict__ opac, uint32_t *__restrict__ tile_box,
                                                              uint2 *__restrict__ depth_pairs)
{
    constexpr int R = (K - 1) * 3;
    __shared__ __attribute__((aligned(16))) float srest[R > 0 ? 256 * R : 4];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int64_t i = i0 + tid;
    const int64_t ic = i < N ? i : N - 1;
    const float p0 = means[3 * ic], p1 = means[3 * ic + 1], p2 = means[3 * ic + 2];
    const float l0 = log_scales[3 * ic], l1 = log_scales[3 * ic + 1], l2 = log_scales[3 * ic + 2];
    float4 q = *reinterpret_cast<const float4 *>(quats + 4 * ic);
    const float opl = op_logit[ic];
    const float d0 = f_dc[3 * ic], d1 = f_dc[3 * ic + 1], d2 = f_dc[3 * ic + 2];
    if (R > 0 && n_use > 0) {
        const int64_t cnt = ((N - i0 < 256 ? N - i0 : 256)) * R;
        const float *src = f_rest + i0 * R;
        for (int64_t j = tid; j < cnt / 4; j += 256) reinterpret_cast<float4 *>(srest)[j] = reinterpret_cast<const float4 *>(src)[j];
        for (int64_t j = (cnt / 4) * 4 + tid; j < cnt; j += 256) srest[j] = src[j];
        __syncthreads();
    }
    if (i >= N) return;
    const float s0 = expf(l0), s1 = expf(l1), s2 = expf(l2);
    const float qn = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    q.x = q.x / qn; q.y = q.y / qn; q.z = q.z / qn; q.w = q.w / qn;
    const float op = sigmoidf(opl);
    opac[i] = op;
    const float *r = srest + tid * R;
    for (int v = 0; v < cb.C; ++v) {
        const Cam &cam = cb.cam[v];
        const int64_t o = (int64_t)v * N + i;
        Proj pr;
        const bool ok = project_one(cam, p0, p1, p2, s0, s1, s2, q.x, q.y, q.z, q.w, pr);
        xys[2 * o] = pr.xy[0]; xys[2 * o + 1] = pr.xy[1];
        if (tile_box) {
            uint32_t box = 0;
            if (ok) box = tight_tile_box(cam, pr, op);
            tile_box[o] = box;
            pr.tiles_hit = (int)(((box >> 8) & 255u) - (box & 255u)) * (int)((box >> 24) - ((box >> 16) & 255u));
        }
        depths[o] = pr.depth; radii[o] = pr.radius; tiles_hit[o] = pr.tiles_hit;
        conics[3 * o] = pr.conic[0]; conics[3 * o + 1] = pr.conic[1]; conics[3 * o + 2] = pr.conic[2];
        if (depth_pairs) depth_pairs[o] = make_uint2(pr.radius > 0 ? __float_as_uint(pr.depth) : 0xFFFFFFFFu, (uint32_t)i);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (ok) {
            if (n_use < 0) {
                c0 = sigmoidf(d0); c1 = sigmoidf(d1); c2 = sigmoidf(d2);
            } else {
                float dx = p0 - cam.ox, dy = p1 - cam.oy, dz = p2 - cam.oz;
                float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
                dx = dx / dn; dy = dy / dn; dz = dz / dn;
                float B[16];
                sh_basis(n_use, dx, dy, dz, B);
                c0 = B[0] * d0; c1 = B[0] * d1; c2 = B[0] * d2;
                int Ku = (n_use + 1) * (n_use + 1);
#pragma unroll
                for (int k = 1; k < K; ++k)
                    if (k < Ku) {
                        c0 += B[k] * r[3 * (k - 1)]; c1 += B[k] * r[3 * (k - 1) + 1]; c2 += B[k] * r[3 * (k - 1) + 2];
                    }
                c0 = fmaxf(c0 + 0.5f, 0.f); c1 = fmaxf(c1 + 0.5f, 0.f); c2 = fmaxf(c2 + 0.5f, 0.f);
            }
        }
        rgbs[3 * o] = c0; rgbs[3 * o + 1] = c1; rgbs[3 * o + 2] = c2;
    }
}

// Backward over C views: the parameter record is read once, the per-view VJPs (same device functions as the single-view kernel) are summed
// in registers IN VIEW ORDER -- ((g_0 + g_1) + g_2) ..., the order C accumulating single-view launches produce -- and the 59 gradient floats
// are written (or, ACC, added to what is there) ONCE per batch: N * (56 + C * 64) bytes read + N * 236 written instead of C * N * 344.
template <int K, bool ACC>
__global__ __launch_bounds__(256) void k_project_sh_bwd_views(int64_t N, CamBatch cb, int n_use,
                                                              const float *__restrict__ means, const float *__restrict__ log_scales,
                                                              const float *__restrict__ quats, const float *__restrict__ op_logit,
                                                              const float *__restrict__ rgbs,
                                                              const int32_t *__restrict__ radii, const float *__restrict__ conics,
                                                              const float *__restrict__ v_xy, const float *__restrict__ v_conic,
                                                              const float *__restrict__ v_rgbs, const float *__restrict__ v_opac,
                                                              float *__restrict__ v_means, float *__restrict__ v_ls,
                                                              float *__restrict__ v_quats, float *__restrict__ v_oplogit,
                                                              float *__restrict__ v_dc, float *__restrict__ v_rest)
{
    constexpr int R = (K - 1) * 3;
    __shared__ __attribute__((aligned(16))) float svr[R > 0 ? 256 * R : 4];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int64_t i = i0 + tid;
    float *vr = svr + tid * R;
    if (i < N) {
        float am[3] = {0.f, 0.f, 0.f}, as[3] = {0.f, 0.f, 0.f}, aq[4] = {0.f, 0.f, 0.f, 0.f}, aop = 0.f, adc[3] = {0.f, 0.f, 0.f};
        // (the 45 features_rest sums live in the lane's own LDS row -- the staging buffer of the coalesced store below; in registers the
        // kernel needs 173 VGPRs = 2 workgroups per CU, with them in LDS the 46 KB of LDS is the limit again: 3 per CU)
#pragma unroll
        for (int k = 0; k < R; ++k) vr[k] = 0.f;
        bool first = true;          // the first contributing view ASSIGNS (0 + g would turn a -0 into +0: keep the single-view bits)
        const float p0 = means[3 * i], p1 = means[3 * i + 1], p2 = means[3 * i + 2];
        const float s0 = expf(log_scales[3 * i]), s1 = expf(log_scales[3 * i + 1]), s2 = expf(log_scales[3 * i + 2]);
        const float4 qr = *reinterpret_cast<const float4 *>(quats + 4 * i);
        const float qn = sqrtf(((qr.x * qr.x + qr.y * qr.y) + qr.z * qr.z) + qr.w * qr.w);
        const float q0 = qr.x / qn, q1 = qr.y / qn, q2 = qr.z / qn, q3 = qr.w / qn;
        const float op = sigmoidf(op_logit[i]);
        for (int v = 0; v < cb.C; ++v) {
            const int64_t o = (int64_t)v * N + i;
            if (radii[o] <= 0) continue;
            const Cam &cam = cb.cam[v];
            ProjGrad g;
            project_one_bwd(cam, p0, p1, p2, s0, s1, s2, q0, q1, q2, q3, conics[3 * o], conics[3 * o + 1], conics[3 * o + 2],
                            v_xy[2 * o], v_xy[2 * o + 1], 0.f, v_conic[3 * o], v_conic[3 * o + 1], v_conic[3 * o + 2], g);
            const float dq = q0 * g.vq[0] + q1 * g.vq[1] + q2 * g.vq[2] + q3 * g.vq[3];
            const float gq[4] = {(g.vq[0] - q0 * dq) / qn, (g.vq[1] - q1 * dq) / qn, (g.vq[2] - q2 * dq) / qn, (g.vq[3] - q3 * dq) / qn};
            const float gls[3] = {g.vs[0] * s0, g.vs[1] * s1, g.vs[2] * s2};
            const float gop = v_opac[o] * op * (1.f - op);
            const float r0 = rgbs[3 * o], r1 = rgbs[3 * o + 1], r2 = rgbs[3 * o + 2];
            float gdc[3], v0 = 0.f, v1 = 0.f, v2 = 0.f;
            float B[16];
            int Ku = 0;
            if (n_use < 0) {
                gdc[0] = v_rgbs[3 * o] * r0 * (1.f - r0); gdc[1] = v_rgbs[3 * o + 1] * r1 * (1.f - r1); gdc[2] = v_rgbs[3 * o + 2] * r2 * (1.f - r2);
#pragma unroll
                for (int k = 0; k < 16; ++k) B[k] = 0.f;
            } else {
                float dx = p0 - cam.ox, dy = p1 - cam.oy, dz = p2 - cam.oz;
                float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
                dx = dx / dn; dy = dy / dn; dz = dz / dn;
                sh_basis(n_use, dx, dy, dz, B);
                Ku = (n_use + 1) * (n_use + 1);
                v0 = r0 > 0.f ? v_rgbs[3 * o] : 0.f;
                v1 = r1 > 0.f ? v_rgbs[3 * o + 1] : 0.f;
                v2 = r2 > 0.f ? v_rgbs[3 * o + 2] : 0.f;
                gdc[0] = B[0] * v0; gdc[1] = B[0] * v1; gdc[2] = B[0] * v2;
            }
            if (first) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { am[k] = g.vm[k]; as[k] = gls[k]; adc[k] = gdc[k]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) aq[k] = gq[k];
                aop = gop;
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    const float b = k < Ku ? B[k] : 0.f;
                    vr[3 * (k - 1)] = b * v0; vr[3 * (k - 1) + 1] = b * v1; vr[3 * (k - 1) + 2] = b * v2;
                }
                first = false;
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) { am[k] = am[k] + g.vm[k]; as[k] = as[k] + gls[k]; adc[k] = adc[k] + gdc[k]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) aq[k] = aq[k] + gq[k];
                aop = aop + gop;
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    const float b = k < Ku ? B[k] : 0.f;
                    vr[3 * (k - 1)] = vr[3 * (k - 1)] + b * v0; vr[3 * (k - 1) + 1] = vr[3 * (k - 1) + 1] + b * v1;
                    vr[3 * (k - 1) + 2] = vr[3 * (k - 1) + 2] + b * v2;
                }
            }
        }
        if (ACC) {
            if (!first) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { v_means[3 * i + k] += am[k]; v_ls[3 * i + k] += as[k]; v_dc[3 * i + k] += adc[k]; }
                float4 o4 = *reinterpret_cast<const float4 *>(v_quats + 4 * i);
                o4.x += aq[0]; o4.y += aq[1]; o4.z += aq[2]; o4.w += aq[3];
                *reinterpret_cast<float4 *>(v_quats + 4 * i) = o4;
                v_oplogit[i] += aop;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { v_means[3 * i + k] = am[k]; v_ls[3 * i + k] = as[k]; v_dc[3 * i + k] = adc[k]; }
            *reinterpret_cast<float4 *>(v_quats + 4 * i) = make_float4(aq[0], aq[1], aq[2], aq[3]);
            v_oplogit[i] = aop;
        }
    }
    if (R > 0) {
        __syncthreads();
        const int64_t cnt = ((N - i0 < 256 ? N - i0 : 256)) * R;
        float *dst = v_rest + i0 * R;
        for (int64_t j = tid; j < cnt / 4; j += 256) {
            float4 v = reinterpret_cast<const float4 *>(svr)[j];
            if (ACC) { const float4 o = reinterpret_cast<const float4 *>(dst)[j]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            reinterpret_cast<float4 *>(dst)[j] = v;
        }
        for (int64_t j = (cnt / 4) * 4 + tid; j < cnt; j += 256) dst[j] = ACC ? dst[j] + svr[j] : svr[j];
    }
}

// img_out == img_raw: in place.  Otherwise the un-clamped image stays where the compositing wrote it (the backward needs it for the
// clamp's gradient mask) and the clamped one goes to img_out: no separate copy pass.
__global__ __launch_bounds__(256) void k_finalize(int64_t npix, const float *img_raw, float *img_out, float *__restrict__ extra,
                                                  const float *__restrict__ final_T, float *__restrict__ alpha)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float a = 1.f - final_T[i];
    alpha[i] = a;
    const float r = img_raw[3 * i], g = img_raw[3 * i + 1], b = img_raw[3 * i + 2];
    img_out[3 * i] = fminf(r, 1.f); img_out[3 * i + 1] = fminf(g, 1.f); img_out[3 * i + 2] = fminf(b, 1.f);
    if (extra) extra[i] = a > 0.f ? extra[i] / a : 1000.f;
}

Cam make_cam(const float *viewmat, const float *projmat, float fx, float fy, float cx, float cy, int H, int W,
             int tx, int ty, float clip, float glob, const float *origin)
{
    Cam c;
    for (int k = 0; k < 12; ++k) c.V[k] = viewmat[k];
    for (int k = 0; k < 16; ++k) c.P[k] = projmat[k];
    c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy; c.H = H; c.W = W; c.tiles_x = tx; c.tiles_y = ty;
    c.clip = clip; c.glob = glob;
    c.ox = origin ? origin[0] : 0.f; c.oy = origin ? origin[1] : 0.f; c.oz = origin ? origin[2] : 0.f;
    return c;
}

}  // namespace

extern "C" {

int gc_project_gaussians_fwd(int64_t N, const float *means3d, const float *scales, float glob_scale,
                             const float *quats, const float *viewmat, const float *projmat, float fx, float fy,
                             float cx, float cy, int img_h, int img_w, int tiles_x, int tiles_y, float clip_thresh,
                             float *cov3d, float *xys, float *depths, int32_t *radii, float *conics,
                             int32_t *num_tiles_hit, void *stream)
{
    GC_REQUIRE(N >= 0 && viewmat && projmat, "bad arguments");
    if (N == 0) return GC_OK;
    Cam cam = make_cam(viewmat, projmat, fx, fy, cx, cy, img_h, img_w, tiles_x, tiles_y, clip_thresh, glob_scale, nullptr);
    hipLaunchKernelGGL(k_project_fwd, dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), N, cam, means3d, scales,
                       quats, cov3d, xys, depths, radii, conics, num_tiles_hit);
    return gc::check_launch("gc_project_gaussians_fwd");
}

int gc_project_gaussians_bwd(int64_t N, const float *means3d, const float *scales, float glob_scale,
                             const float *quats, const float *viewmat, const float *projmat, float fx, float fy,
                             float cx, float cy, int img_h, int img_w, const int32_t *radii, const float *conics,
                             const float *v_xy, const float *v_depth, const float *v_conic, float *v_mean3d,
                             float *v_scale, float *v_quat, void *stream)
{
    GC_REQUIRE(N >= 0 && viewmat && projmat, "bad arguments");
    if (N == 0) return GC_OK;
    Cam cam = make_cam(viewmat, projmat, fx, fy, cx, cy, img_h, img_w, 0, 0, 0.f, glob_scale, nullptr);
    hipLaunchKernelGGL(k_project_bwd, dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), N, cam, means3d, scales,
                       quats, radii, conics, v_xy, v_depth, v_conic, v_mean3d, v_scale, v_quat);
    return gc::check_launch("gc_project_gaussians_bwd");
}

int gc_sh_fwd(int64_t N, int degree, int degrees_to_use, const float *viewdirs, const float *coeffs, float *colors,
              void *stream)
{
    GC_REQUIRE(degree >= 0 && degree <= 3 && degrees_to_use >= 0 && degrees_to_use <= degree, "SH degree must be 0..3");
    if (N == 0) return GC_OK;
    int K = (degree + 1) * (degree + 1);
    hipLaunchKernelGGL(k_sh_fwd, dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), N, K, degrees_to_use, viewdirs,
                       coeffs, colors);
    return gc::check_launch("gc_sh_fwd");
}

int gc_sh_bwd(int64_t N, int degree, int degrees_to_use, const float *viewdirs, const float *v_colors,
              float *v_coeffs, void *stream)
{
    GC_REQUIRE(degree >= 0 && degree <= 3 && degrees_to_use >= 0 && degrees_to_use <= degree, "SH degree must be 0..3");
    if (N == 0) return GC_OK;
    int K = (degree + 1) * (degree + 1);
    hipLaunchKernelGGL(k_sh_bwd, dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), N, K, degrees_to_use, viewdirs,
                       v_colors, v_coeffs);
    return gc::check_launch("gc_sh_bwd");
}

#define GC_SH_DISPATCH(KERNEL, ...)                                                                               \
    switch (sh_degree) {                                                                                          \
    case 0: hipLaunchKernelGGL(KERNEL<1>, dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), __VA_ARGS__); break; \
    case 1: hipLaunchKernelGGL(KERNEL<4>, dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), __VA_ARGS__); break; \
    case 2: hipLaunchKernelGGL(KERNEL<9>, dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), __VA_ARGS__); break; \
    default: hipLaunchKernelGGL(KERNEL<16>, dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), __VA_ARGS__); break; \
    }

static int project_sh_fwd_impl(const char *what, int64_t N, const float *means, const float *log_scales, const float *quats,
                      const float *opacity_logits, const float *features_dc, const float *features_rest,
                      int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                      const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                      int tiles_x, int tiles_y, float clip_thresh, float *xys, float *depths, int32_t *radii,
                      float *conics, int32_t *num_tiles_hit, float *rgbs, float *opac, uint32_t *tile_boxes, void *stream)
{
    GC_REQUIRE(sh_degree >= 0 && sh_degree <= 3 && degrees_to_use >= -1 && degrees_to_use <= sh_degree, "SH degree must be 0..3 (degrees_to_use -1: sigmoid colour mode)");
    GC_REQUIRE(viewmat && projmat && cam_origin, "camera pointers are host pointers and must not be NULL");
    if (N == 0) return GC_OK;
    Cam cam = make_cam(viewmat, projmat, fx, fy, cx, cy, img_h, img_w, tiles_x, tiles_y, clip_thresh, 1.f, cam_origin);
    GC_SH_DISPATCH(k_project_sh_fwd, N, cam, degrees_to_use, means, log_scales, quats, opacity_logits, features_dc,
                   features_rest, xys, depths, radii, conics, num_tiles_hit, rgbs, opac, tile_boxes)
    return gc::check_launch(what);
}

int gc_project_sh_fwd(int64_t N, const float *means, const float *log_scales, const float *quats,
                      const float *opacity_logits, const float *features_dc, const float *features_rest,
                      int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                      const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                      int tiles_x, int tiles_y, float clip_thresh, float *xys, float *depths, int32_t *radii,
                      float *conics, int32_t *num_tiles_hit, float *rgbs, float *opac, void *stream)
{
    return project_sh_fwd_impl("gc_project_sh_fwd", N, means, log_scales, quats, opacity_logits, features_dc, features_rest, sh_degree, degrees_to_use,
                               viewmat, projmat, cam_origin, fx, fy, cx, cy, img_h, img_w, tiles_x, tiles_y, clip_thresh, xys, depths, radii, conics,
                               num_tiles_hit, rgbs, opac, nullptr, stream);
}

/* The same with TIGHT tile boxes: tile_boxes[N] (packed, see tight_tile_box) and num_tiles_hit count only the tiles of the bounding box
 * of the alpha >= 1/255 ellipse inside gsplat's box; feed both to gc_raster_depth_order / gc_raster_bin_tiles_boxes. */
int gc_project_sh_fwd_boxes(int64_t N, const float *means, const float *log_scales, const float *quats,
                            const float *opacity_logits, const float *features_dc, const float *features_rest,
                            int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                            const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                            int tiles_x, int tiles_y, float clip_thresh, float *xys, float *depths, int32_t *radii,
                            float *conics, int32_t *num_tiles_hit, float *rgbs, float *opac, uint32_t *tile_boxes, void *stream)
{
    GC_REQUIRE(tile_boxes && tiles_x <= 255 && tiles_y <= 255, "tile_boxes required; packed boxes hold at most 255 x 255 tiles (use gc_project_sh_fwd beyond)");
    return project_sh_fwd_impl("gc_project_sh_fwd_boxes", N, means, log_scales, quats, opacity_logits, features_dc, features_rest, sh_degree, degrees_to_use,
                               viewmat, projmat, cam_origin, fx, fy, cx, cy, img_h, img_w, tiles_x, tiles_y, clip_thresh, xys, depths, radii, conics,
                               num_tiles_hit, rgbs, opac, tile_boxes, stream);
}

static int project_sh_bwd_impl(bool accumulate, int64_t N, const float *means, const float *log_scales, const float *quats,
                      const float *opacity_logits, const float *rgbs,
                      int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                      const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                      const int32_t *radii, const float *conics, const float *v_xy, const float *v_conic,
                      const float *v_rgbs, const float *v_opac, float *v_means, float *v_log_scales, float *v_quats,
                      float *v_opacity_logits, float *v_features_dc, float *v_features_rest, void *stream)
{
    GC_REQUIRE(sh_degree >= 0 && sh_degree <= 3 && degrees_to_use >= -1 && degrees_to_use <= sh_degree, "SH degree must be 0..3 (degrees_to_use -1: sigmoid colour mode)");
    GC_REQUIRE(viewmat && projmat && cam_origin, "camera pointers are host pointers and must not be NULL");
    if (N == 0) return GC_OK;
    Cam cam = make_cam(viewmat, projmat, fx, fy, cx, cy, img_h, img_w, 0, 0, 0.f, 1.f, cam_origin);
#define GC_BWD_K(KK) \
    do { if (accumulate) hipLaunchKernelGGL((k_project_sh_bwd<KK, true>), dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), GC_BWD_ARGS); \
         else hipLaunchKernelGGL((k_project_sh_bwd<KK, false>), dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), GC_BWD_ARGS); } while (0)
#define GC_BWD_ARGS N, cam, degrees_to_use, means, log_scales, quats, opacity_logits, rgbs, radii, conics, v_xy, v_conic, v_rgbs, v_opac, v_means, v_log_scales, v_quats, v_opacity_logits, v_features_dc, v_features_rest
    switch (sh_degree) { case 0: GC_BWD_K(1); break; case 1: GC_BWD_K(4); break; case 2: GC_BWD_K(9); break; default: GC_BWD_K(16); break; }
#undef GC_BWD_K
#undef GC_BWD_ARGS
    return gc::check_launch("gc_project_sh_bwd");
}

int gc_project_sh_bwd(int64_t N, const float *means, const float *log_scales, const float *quats,
                      const float *opacity_logits, const float *rgbs,
                      int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                      const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                      const int32_t *radii, const float *conics, const float *v_xy, const float *v_conic,
                      const float *v_rgbs, const float *v_opac, float *v_means, float *v_log_scales, float *v_quats,
                      float *v_opacity_logits, float *v_features_dc, float *v_features_rest, void *stream)
{
    return project_sh_bwd_impl(false, N, means, log_scales, quats, opacity_logits, rgbs, sh_degree, degrees_to_use, viewmat, projmat, cam_origin, fx, fy, cx, cy, img_h, img_w, radii, conics, v_xy, v_conic, v_rgbs, v_opac, v_means, v_log_scales, v_quats, v_opacity_logits, v_features_dc, v_features_rest, stream);
}

/* Same, but the six outputs are ACCUMULATED into (+=): gradient accumulation over the views of a batch inside the kernel. */
int gc_project_sh_bwd_accumulate(int64_t N, const float *means, const float *log_scales, const float *quats,
                      const float *opacity_logits, const float *rgbs,
                      int sh_degree, int degrees_to_use, const float *viewmat, const float *projmat,
                      const float *cam_origin, float fx, float fy, float cx, float cy, int img_h, int img_w,
                      const int32_t *radii, const float *conics, const float *v_xy, const float *v_conic,
                      const float *v_rgbs, const float *v_opac, float *v_means, float *v_log_scales, float *v_quats,
                      float *v_opacity_logits, float *v_features_dc, float *v_features_rest, void *stream)
{
    return project_sh_bwd_impl(true, N, means, log_scales, quats, opacity_logits, rgbs, sh_degree, degrees_to_use, viewmat, projmat, cam_origin, fx, fy, cx, cy, img_h, img_w, radii, conics, v_xy, v_conic, v_rgbs, v_opac, v_means, v_log_scales, v_quats, v_opacity_logits, v_features_dc, v_features_rest, stream);
}

/* ---- C views per launch (round 5).  cams: HOST float array [C][GC_VIEW_CAM_FLOATS = 35] = viewmat[12] | projmat[16] | cam_origin[3] |
 * fx fy cx cy per view; all views share H, W and the tile grid.  Views are processed in groups of 8 (one launch per group: the cameras
 * travel as kernel arguments); outputs are [C][N][..] except opac [N].  tile_boxes / depth_pairs optional (NULL). */
static int fill_cams(CamBatch &cb, const float *cams, int v0, int nv, int img_h, int img_w, int tx, int ty, float clip)
{
    cb.C = nv;
    for (int v = 0; v < nv; ++v) {
        const float *c = cams + (size_t)(v0 + v) * 35;
        cb.cam[v] = make_cam(c, c + 12, c[31], c[32], c[33], c[34], img_h, img_w, tx, ty, clip, 1.f, c + 28);
    }
    return 0;
}

int gc_project_sh_fwd_views(int64_t N, int C, const float *means, const float *log_scales, const float *quats,
                            const float *opacity_logits, const float *features_dc, const float *features_rest,
                            int sh_degree, int degrees_to_use, const float *cams, int img_h, int img_w,
                            int tiles_x, int tiles_y, float clip_thresh, float *xys, float *depths, int32_t *radii,
                            float *conics, int32_t *num_tiles_hit, float *rgbs, float *opac, uint32_t *tile_boxes,
                            uint32_t *depth_pairs, void *stream)
{
    GC_REQUIRE(sh_degree >= 0 && sh_degree <= 3 && degrees_to_use >= -1 && degrees_to_use <= sh_degree, "SH degree must be 0..3 (degrees_to_use -1: sigmoid colour mode)");
    GC_REQUIRE(cams && C >= 1, "cams is a host pointer and must not be NULL");
    GC_REQUIRE(!tile_boxes || (tiles_x <= 255 && tiles_y <= 255), "packed boxes hold at most 255 x 255 tiles");
    if (N == 0) return GC_OK;
    for (int v0 = 0; v0 < C; v0 += MAXV) {
        CamBatch cb;
        fill_cams(cb, cams, v0, C - v0 < MAXV ? C - v0 : MAXV, img_h, img_w, tiles_x, tiles_y, clip_thresh);
        const size_t o = (size_t)v0 * (size_t)N;
        GC_SH_DISPATCH(k_project_sh_fwd_views, N, cb, degrees_to_use, means, log_scales, quats, opacity_logits, features_dc, features_rest,
                       xys + 2 * o, depths + o, radii + o, conics + 3 * o, num_tiles_hit + o, rgbs + 3 * o, opac,
                       tile_boxes ? tile_boxes + o : nullptr, depth_pairs ? (uint2 *)depth_pairs + o : nullptr)
    }
    return gc::check_launch("gc_project_sh_fwd_views");
}

/* Backward over C views: rgbs / radii / conics / v_xy / v_conic / v_rgbs / v_opac are [C][N][..]; the six leaf gradients are the SUM over
 * the views, written (accumulate = 0) or added to the buffers' contents (accumulate = 1) once per group of 8 views. */
int gc_project_sh_bwd_views(int64_t N, int C, int accumulate, const float *means, const float *log_scales, const float *quats,
                            const float *opacity_logits, const float *rgbs, int sh_degree, int degrees_to_use, const float *cams,
                            int img_h, int img_w, const int32_t *radii, const float *conics, const float *v_xy, const float *v_conic,
                            const float *v_rgbs, const float *v_opac, float *v_means, float *v_log_scales, float *v_quats,
                            float *v_opacity_logits, float *v_features_dc, float *v_features_rest, void *stream)
{
    GC_REQUIRE(sh_degree >= 0 && sh_degree <= 3 && degrees_to_use >= -1 && degrees_to_use <= sh_degree, "SH degree must be 0..3 (degrees_to_use -1: sigmoid colour mode)");
    GC_REQUIRE(cams && C >= 1, "cams is a host pointer and must not be NULL");
    if (N == 0) return GC_OK;
    for (int v0 = 0; v0 < C; v0 += MAXV) {
        CamBatch cb;
        fill_cams(cb, cams, v0, C - v0 < MAXV ? C - v0 : MAXV, img_h, img_w, 0, 0, 0.f);
        const size_t o = (size_t)v0 * (size_t)N;
        const bool acc = accumulate || v0 > 0;
#define GC_BWDV_ARGS N, cb, degrees_to_use, means, log_scales, quats, opacity_logits, rgbs + 3 * o, radii + o, conics + 3 * o, v_xy + 2 * o, v_conic + 3 * o, v_rgbs + 3 * o, v_opac + o, v_means, v_log_scales, v_quats, v_opacity_logits, v_features_dc, v_features_rest
#define GC_BWDV_K(KK) \
        do { if (acc) hipLaunchKernelGGL((k_project_sh_bwd_views<KK, true>), dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), GC_BWDV_ARGS); \
             else hipLaunchKernelGGL((k_project_sh_bwd_views<KK, false>), dim3(gc::cdiv(N, 256)), dim3(256), 0, gc::S(stream), GC_BWDV_ARGS); } while (0)
        switch (sh_degree) { case 0: GC_BWDV_K(1); break; case 1: GC_BWDV_K(4); break; case 2: GC_BWDV_K(9); break; default: GC_BWDV_K(16); break; }
#undef GC_BWDV_K
#undef GC_BWDV_ARGS
    }
    return gc::check_launch("gc_project_sh_bwd_views");
}

int gc_raster_finalize(int64_t num_pixels, float *out_img, float *out_extra, const float *final_Ts, float *alpha,
                       void *stream)
{
    if (num_pixels == 0) return GC_OK;
    hipLaunchKernelGGL(k_finalize, dim3(gc::cdiv(num_pixels, 256)), dim3(256), 0, gc::S(stream), num_pixels, (const float *)out_img, out_img,
                       out_extra, final_Ts, alpha);
    return gc::check_launch("gc_raster_finalize");
}

int gc_raster_finalize_into(int64_t num_pixels, const float *img_raw, float *img_clamped, float *out_extra, const float *final_Ts,
                            float *alpha, void *stream)
{
    if (num_pixels == 0) return GC_OK;
    GC_REQUIRE(img_raw && img_clamped && img_raw != img_clamped, "needs two distinct image buffers (gc_raster_finalize is the in-place form)");
    hipLaunchKernelGGL(k_finalize, dim3(gc::cdiv(num_pixels, 256)), dim3(256), 0, gc::S(stream), num_pixels, img_raw, img_clamped,
                       out_extra, final_Ts, alpha);
    return gc::check_launch("gc_raster_finalize_into");
}

}  // extern "C"
