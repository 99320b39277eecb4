/*
 * oracle/raster_ref.c  --  TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never measured
 * as the product).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * Plain-C (fp32) restatement of the 3D-Gaussian rasterizer semantics the reference reaches
 * through gsplat 0.1.3 (third-party, NOT under /root/reference; pinned in prose only at
 * /root/reference/README.md:53-60).  Reference call sites this restates:
 *   project_gaussians    /root/reference/gaussctrl/gc_model.py:140-154
 *   spherical_harmonics  /root/reference/gaussctrl/gc_model.py:166
 *   rasterize_gaussians  /root/reference/gaussctrl/gc_model.py:174-186 and :191-202
 *   backward of the three, fired by /root/reference/gaussctrl/gc_trainer.py:275
 * Algorithm spec: SURVEY.md Appendix A (A.1 project, A.2 SH, A.3 bin&sort, A.4 rasterize
 * forward, A.5 backward).
 *
 * PARITY UNPINNED at the gsplat boundary: the reference repo holds no golden vectors, tests or
 * fixtures for this path and gsplat cannot be built or imported here (CUDA only).  The oracle is
 * pinned instead by (i) analytic known-answer tests (tests/test_oracle_raster.py) and (ii) fp64
 * torch autograd of an independent vectorised restatement (oracle/raster_torch.py).
 *
 * Floating-point contract: every expression below is written as explicit IEEE-754 binary32
 * add/mul/div/sqrt in a fixed order; compile with -ffp-contract=off and WITHOUT -ffast-math so
 * that the HIP projection kernel (which mirrors the order) produces bit-identical integer
 * outputs (radii, tile boxes, num_tiles_hit, sort keys).
 *
 * Backward convention: gradients are the TRUE derivatives of the forward map (what autograd of
 * the forward gives), including d(conic.y) = dx*dy*v_sigma and zero gradient through the
 * alpha cap; gsplat's internal 0.5 factor on the off-diagonal conic gradient is not observable
 * through the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Optional threaded build (liboracle_raster_mt.so: -fopenmp -DORC_MT), used ONLY by bench.py's cpu_baseline leg so that the CPU
 * figure uses every host core.  The default build (what the parity tests load) stays single-threaded and deterministic: without
 * ORC_MT the macros below expand to the plain serial statements. */
#ifdef ORC_MT
#define ORC_PAR _Pragma("omp parallel for schedule(dynamic, 256)")
#define ORC_PAR_ROWS _Pragma("omp parallel for schedule(dynamic, 1)")
#define ORC_ADD(dst, v) do { _Pragma("omp atomic") dst += (v); } while (0)
#else
#define ORC_PAR
#define ORC_PAR_ROWS
#define ORC_ADD(dst, v) dst += (v)
#endif

#define TILE 16
#define ALPHA_CAP 0.999f
#define ALPHA_MIN (1.f / 255.f)
#define T_STOP 1e-4f

/* ---------------------------------------------------------------- A.1 projection forward -- */

static void quat_to_rotmat(const float *q, float *R /* row-major 3x3 */, float *qn /* 4 */, float *inv_norm)
{
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float n2 = ((w * w + x * x) + y * y) + z * z;
    float s = 1.f / sqrtf(n2);
    w = w * s; x = x * s; y = y * s; z = z * s;
    if (qn) { qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z; }
    if (inv_norm) *inv_norm = s;
    R[0] = 1.f - 2.f * (y * y + z * z);
    R[1] = 2.f * (x * y - w * z);
    R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z);
    R[4] = 1.f - 2.f * (x * x + z * z);
    R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y);
    R[7] = 2.f * (y * z + w * x);
    R[8] = 1.f - 2.f * (x * x + y * y);
}

/* Sigma = M M^T with M = R diag(g*s); upper triangle, 6 floats. */
static void scale_rot_to_cov3d(const float *scale, float glob, const float *R, float *cov3d, float *M)
{
    float sx = glob * scale[0], sy = glob * scale[1], sz = glob * scale[2];
    float m[9];
    m[0] = R[0] * sx; m[1] = R[1] * sy; m[2] = R[2] * sz;
    m[3] = R[3] * sx; m[4] = R[4] * sy; m[5] = R[5] * sz;
    m[6] = R[6] * sx; m[7] = R[7] * sy; m[8] = R[8] * sz;
    cov3d[0] = (m[0] * m[0] + m[1] * m[1]) + m[2] * m[2];
    cov3d[1] = (m[0] * m[3] + m[1] * m[4]) + m[2] * m[5];
    cov3d[2] = (m[0] * m[6] + m[1] * m[7]) + m[2] * m[8];
    cov3d[3] = (m[3] * m[3] + m[4] * m[4]) + m[5] * m[5];
    cov3d[4] = (m[3] * m[6] + m[4] * m[7]) + m[5] * m[8];
    cov3d[5] = (m[6] * m[6] + m[7] * m[7]) + m[8] * m[8];
    if (M) memcpy(M, m, sizeof(m));
}

static inline float clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* returns 1 if the Gaussian survives; fills the per-Gaussian outputs */
static int project_one(const float *p, const float *scale, float glob, const float *quat,
                       const float *V /* >=12, row-major 3x4 */, const float *P /* 16 */,
                       float fx, float fy, float cx, float cy, int H, int W,
                       int tiles_x, int tiles_y, float clip,
                       float *cov3d, float *xy, float *depth, int32_t *radius_out, float *conic,
                       int32_t *tiles_hit)
{
    *radius_out = 0; *tiles_hit = 0;
    float tx = ((V[0] * p[0] + V[1] * p[1]) + V[2] * p[2]) + V[3];
    float ty = ((V[4] * p[0] + V[5] * p[1]) + V[6] * p[2]) + V[7];
    float tz = ((V[8] * p[0] + V[9] * p[1]) + V[10] * p[2]) + V[11];
    if (tz <= clip) return 0;

    float R[9];
    quat_to_rotmat(quat, R, NULL, NULL);
    scale_rot_to_cov3d(scale, glob, R, cov3d, NULL);

    /* EWA */
    float lim_x = 1.3f * (0.5f * (float)W / fx);
    float lim_y = 1.3f * (0.5f * (float)H / fy);
    float txc = tz * clampf(tx / tz, -lim_x, lim_x);
    float tyc = tz * clampf(ty / tz, -lim_y, lim_y);
    float rz = 1.f / tz;
    float rz2 = rz * rz;
    float j00 = fx * rz, j02 = -(fx * txc) * rz2;
    float j11 = fy * rz, j12 = -(fy * tyc) * rz2;
    /* T = J W3 (2x3) */
    float t00 = j00 * V[0] + j02 * V[8], t01 = j00 * V[1] + j02 * V[9], t02 = j00 * V[2] + j02 * V[10];
    float t10 = j11 * V[4] + j12 * V[8], t11 = j11 * V[5] + j12 * V[9], t12 = j11 * V[6] + j12 * V[10];
    /* U = T Sigma (2x3) */
    const float *c = cov3d;
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
    if (det == 0.f) return 0;
    float inv_det = 1.f / det;
    conic[0] = d * inv_det;
    conic[1] = -b * inv_det;
    conic[2] = a * inv_det;
    float mid = 0.5f * (a + d);
    float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    float v1 = mid + disc, v2 = mid - disc;
    float radius = ceilf(3.f * sqrtf(fmaxf(v1, v2)));

    /* pixel centre through the full projection */
    float hx = ((P[0] * p[0] + P[1] * p[1]) + P[2] * p[2]) + P[3];
    float hy = ((P[4] * p[0] + P[5] * p[1]) + P[6] * p[2]) + P[7];
    float hw = ((P[12] * p[0] + P[13] * p[1]) + P[14] * p[2]) + P[15];
    float rw = 1.f / (hw + 1e-6f);
    float px = (0.5f * (float)W) * (hx * rw) + cx - 0.5f;
    float py = (0.5f * (float)H) * (hy * rw) + cy - 0.5f;

    /* tile bbox: centre and radius in tile units, C float->int truncation */
    float tcx = px / (float)TILE, tcy = py / (float)TILE, tr = radius / (float)TILE;
    int minx = clampi((int)(tcx - tr), 0, tiles_x), maxx = clampi((int)(tcx + tr + 1.f), 0, tiles_x);
    int miny = clampi((int)(tcy - tr), 0, tiles_y), maxy = clampi((int)(tcy + tr + 1.f), 0, tiles_y);
    int area = (maxx - minx) * (maxy - miny);
    if (area <= 0) return 0;
    *tiles_hit = area;
    *depth = tz;
    *radius_out = (int32_t)radius;
    xy[0] = px; xy[1] = py;
    return 1;
}

void orc_project_gaussians_fwd(int64_t N, const float *means3d, const float *scales, float glob_scale,
                               const float *quats, const float *viewmat, const float *projmat,
                               float fx, float fy, float cx, float cy, int H, int W,
                               int tiles_x, int tiles_y, float clip_thresh,
                               float *cov3d, float *xys, float *depths, int32_t *radii,
                               float *conics, int32_t *num_tiles_hit)
{
    ORC_PAR
    for (int64_t i = 0; i < N; ++i) {
        float c3[6] = {0}, xy[2] = {0}, dep = 0.f, con[3] = {0};
        int32_t rad, nth;
        project_one(means3d + 3 * i, scales + 3 * i, glob_scale, quats + 4 * i, viewmat, projmat,
                    fx, fy, cx, cy, H, W, tiles_x, tiles_y, clip_thresh, c3, xy, &dep, &rad, con, &nth);
        memcpy(cov3d + 6 * i, c3, sizeof(c3));
        xys[2 * i] = xy[0]; xys[2 * i + 1] = xy[1];
        depths[i] = dep; radii[i] = rad; num_tiles_hit[i] = nth;
        conics[3 * i] = con[0]; conics[3 * i + 1] = con[1]; conics[3 * i + 2] = con[2];
    }
}

/* ---------------------------------------------------------------- A.1 projection backward -- */
/* v_xy[N,2], v_depth[N], v_conic[N,3] -> v_mean3d[N,3], v_scale[N,3], v_quat[N,4] (true VJP). */
void orc_project_gaussians_bwd(int64_t N, const float *means3d, const float *scales, float glob_scale,
                               const float *quats, const float *viewmat, const float *projmat,
                               float fx, float fy, float cx, float cy, int H, int W,
                               const int32_t *radii, const float *conics,
                               const float *v_xy, const float *v_depth, const float *v_conic,
                               float *v_mean3d, float *v_scale, float *v_quat)
{
    (void)cx; (void)cy;
    const float *V = viewmat, *P = projmat;
    ORC_PAR
    for (int64_t i = 0; i < N; ++i) {
        float *vm = v_mean3d + 3 * i, *vs = v_scale + 3 * i, *vq = v_quat + 4 * i;
        vm[0] = vm[1] = vm[2] = 0.f; vs[0] = vs[1] = vs[2] = 0.f; vq[0] = vq[1] = vq[2] = vq[3] = 0.f;
        if (radii[i] <= 0) continue;
        const float *p = means3d + 3 * i;
        /* (a) pixel centre */
        float hx = ((P[0] * p[0] + P[1] * p[1]) + P[2] * p[2]) + P[3];
        float hy = ((P[4] * p[0] + P[5] * p[1]) + P[6] * p[2]) + P[7];
        float hw = ((P[12] * p[0] + P[13] * p[1]) + P[14] * p[2]) + P[15];
        float rw = 1.f / (hw + 1e-6f);
        float vnx = 0.5f * (float)W * v_xy[2 * i], vny = 0.5f * (float)H * v_xy[2 * i + 1];
        float vhx = vnx * rw, vhy = vny * rw, vhw = -(vnx * hx + vny * hy) * rw * rw;
        vm[0] += P[0] * vhx + P[4] * vhy + P[12] * vhw;
        vm[1] += P[1] * vhx + P[5] * vhy + P[13] * vhw;
        vm[2] += P[2] * vhx + P[6] * vhy + P[14] * vhw;
        /* (b) depth */
        float vz = v_depth ? v_depth[i] : 0.f;
        vm[0] += V[8] * vz; vm[1] += V[9] * vz; vm[2] += V[10] * vz;
        /* (c) conic -> cov2d : vC = -X G X, X = conic matrix */
        float X00 = conics[3 * i], X01 = conics[3 * i + 1], X11 = conics[3 * i + 2];
        float G00 = v_conic[3 * i], G01 = 0.5f * v_conic[3 * i + 1], G11 = v_conic[3 * i + 2];
        /* XG */
        float a00 = X00 * G00 + X01 * G01, a01 = X00 * G01 + X01 * G11;
        float a10 = X01 * G00 + X11 * G01, a11 = X01 * G01 + X11 * G11;
        float C00 = -(a00 * X00 + a01 * X01), C01 = -(a00 * X01 + a01 * X11);
        float C11 = -(a10 * X01 + a11 * X11);
        /* matrix-form gradient of the symmetric 2x2 cov: Gc = [[C00, C01],[C01, C11]] */
        /* recompute forward intermediates */
        float tx = ((V[0] * p[0] + V[1] * p[1]) + V[2] * p[2]) + V[3];
        float ty = ((V[4] * p[0] + V[5] * p[1]) + V[6] * p[2]) + V[7];
        float tz = ((V[8] * p[0] + V[9] * p[1]) + V[10] * p[2]) + V[11];
        float lim_x = 1.3f * (0.5f * (float)W / fx), lim_y = 1.3f * (0.5f * (float)H / fy);
        float rx = tx / tz, ry = ty / tz;
        float rxc = clampf(rx, -lim_x, lim_x), ryc = clampf(ry, -lim_y, lim_y);
        int clx = (rx != rxc), cly = (ry != ryc);
        float txc = tz * rxc, tyc = tz * ryc;
        float rz = 1.f / tz, rz2 = rz * rz;
        float j00 = fx * rz, j02 = -(fx * txc) * rz2, j11 = fy * rz, j12 = -(fy * tyc) * rz2;
        float T[6] = { j00 * V[0] + j02 * V[8], j00 * V[1] + j02 * V[9], j00 * V[2] + j02 * V[10],
                       j11 * V[4] + j12 * V[8], j11 * V[5] + j12 * V[9], j11 * V[6] + j12 * V[10] };
        float R[9], qn[4], inv_norm, M[9], c3[6];
        quat_to_rotmat(quats + 4 * i, R, qn, &inv_norm);
        scale_rot_to_cov3d(scales + 3 * i, glob_scale, R, c3, M);
        float S[9] = { c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5] };
        /* (d) vSigma = T^T Gc T (3x3 sym), vT = 2 Gc T Sigma (2x3) */
        float GT[6]; /* Gc T */
        for (int k = 0; k < 3; ++k) { GT[k] = C00 * T[k] + C01 * T[3 + k]; GT[3 + k] = C01 * T[k] + C11 * T[3 + k]; }
        float vS[9];
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) vS[3 * r + k] = T[r] * GT[k] + T[3 + r] * GT[3 + k];
        float vT[6];
        for (int r = 0; r < 2; ++r) for (int k = 0; k < 3; ++k)
            vT[3 * r + k] = 2.f * (GT[3 * r] * S[k] + GT[3 * r + 1] * S[3 + k] + GT[3 * r + 2] * S[6 + k]);
        /* vM = 2 vS M */
        float vM[9];
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k)
            vM[3 * r + k] = 2.f * (vS[3 * r] * M[k] + vS[3 * r + 1] * M[3 + k] + vS[3 * r + 2] * M[6 + k]);
        const float *sc = scales + 3 * i;
        float vR[9];
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) vR[3 * r + k] = vM[3 * r + k] * (glob_scale * sc[k]);
        for (int k = 0; k < 3; ++k) vs[k] = glob_scale * (R[k] * vM[k] + R[3 + k] * vM[3 + k] + R[6 + k] * vM[6 + k]);
        /* R(q^) -> v_qn */
        float w = qn[0], x = qn[1], y = qn[2], z = qn[3];
        float vqn[4];
        vqn[0] = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
        vqn[1] = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
        vqn[2] = 2.f * (x * (vR[1] + vR[3]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
        vqn[3] = 2.f * (x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
        float dotq = qn[0] * vqn[0] + qn[1] * vqn[1] + qn[2] * vqn[2] + qn[3] * vqn[3];
        for (int k = 0; k < 4; ++k) vq[k] = (vqn[k] - qn[k] * dotq) * inv_norm;
        /* (e) T = J W3 -> vJ = vT W3^T */
        float vj00 = vT[0] * V[0] + vT[1] * V[1] + vT[2] * V[2];
        float vj02 = vT[0] * V[8] + vT[1] * V[9] + vT[2] * V[10];
        float vj11 = vT[3] * V[4] + vT[4] * V[5] + vT[5] * V[6];
        float vj12 = vT[3] * V[8] + vT[4] * V[9] + vT[5] * V[10];
        float v_rz = fx * vj00 + fy * vj11 - 2.f * rz * (fx * txc * vj02 + fy * tyc * vj12);
        float v_txc = -fx * rz2 * vj02, v_tyc = -fy * rz2 * vj12;
        float v_tz = -rz2 * v_rz, v_tx = 0.f, v_ty = 0.f;
        if (clx) v_tz += rxc * v_txc; else v_tx += v_txc;
        if (cly) v_tz += ryc * v_tyc; else v_ty += v_tyc;
        vm[0] += V[0] * v_tx + V[4] * v_ty + V[8] * v_tz;
        vm[1] += V[1] * v_tx + V[5] * v_ty + V[9] * v_tz;
        vm[2] += V[2] * v_tx + V[6] * v_ty + V[10] * v_tz;
    }
}

/* ---------------------------------------------------------------- A.2 spherical harmonics -- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f };

static int sh_bases(int degree) { return (degree + 1) * (degree + 1); }

static void sh_basis(int n, const float *d, float *B /* 16 */)
{
    for (int k = 0; k < 16; ++k) B[k] = 0.f;
    B[0] = SH_C0;
    if (n < 1) return;
    float x = d[0], y = d[1], z = d[2];
    B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
    if (n < 2) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.f * zz - xx - yy);
    B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
    if (n < 3) return;
    B[9] = SH_C3[0] * y * (3.f * xx - yy);
    B[10] = SH_C3[1] * xy * z;
    B[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
    B[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
    B[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
    B[14] = SH_C3[5] * z * (xx - yy);
    B[15] = SH_C3[6] * x * (xx - 3.f * yy);
}

/* coeffs[N,K,3] with K = (degree+1)^2 stored bases; only the first (n+1)^2 are used */
void orc_sh_fwd(int64_t N, int degree, int degrees_to_use, const float *viewdirs, const float *coeffs, float *colors)
{
    int K = sh_bases(degree), Ku = sh_bases(degrees_to_use);
    ORC_PAR
    for (int64_t i = 0; i < N; ++i) {
        float B[16];
        sh_basis(degrees_to_use, viewdirs + 3 * i, B);
        for (int c = 0; c < 3; ++c) {
            float acc = 0.f;
            for (int k = 0; k < Ku; ++k) acc += B[k] * coeffs[(i * K + k) * 3 + c];
            colors[3 * i + c] = acc;
        }
    }
}

void orc_sh_bwd(int64_t N, int degree, int degrees_to_use, const float *viewdirs, const float *v_colors, float *v_coeffs)
{
    int K = sh_bases(degree), Ku = sh_bases(degrees_to_use);
    ORC_PAR
    for (int64_t i = 0; i < N; ++i) {
        float B[16];
        sh_basis(degrees_to_use, viewdirs + 3 * i, B);
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < 3; ++c)
                v_coeffs[(i * K + k) * 3 + c] = k < Ku ? B[k] * v_colors[3 * i + c] : 0.f;
    }
}

/* ---------------------------------------------------------------- A.3 bin & sort ---------- */
/* cum[i] = inclusive prefix sum of num_tiles_hit; returns M */
int64_t orc_cumsum_tiles(int64_t N, const int32_t *num_tiles_hit, int32_t *cum)
{
    int32_t acc = 0;
    for (int64_t i = 0; i < N; ++i) { acc += num_tiles_hit[i]; cum[i] = acc; }
    return (int64_t)acc;
}

void orc_map_gaussian_to_intersects(int64_t N, const float *xys, const float *depths, const int32_t *radii,
                                    const int32_t *cum, int tiles_x, int tiles_y,
                                    int64_t *isect_ids, int32_t *gaussian_ids)
{
    for (int64_t i = 0; i < N; ++i) {
        if (radii[i] <= 0) continue;
        float tcx = xys[2 * i] / (float)TILE, tcy = xys[2 * i + 1] / (float)TILE, tr = (float)radii[i] / (float)TILE;
        int minx = clampi((int)(tcx - tr), 0, tiles_x), maxx = clampi((int)(tcx + tr + 1.f), 0, tiles_x);
        int miny = clampi((int)(tcy - tr), 0, tiles_y), maxy = clampi((int)(tcy + tr + 1.f), 0, tiles_y);
        int64_t cur = (i == 0) ? 0 : cum[i - 1];
        int32_t dbits; memcpy(&dbits, depths + i, 4);
        for (int ty = miny; ty < maxy; ++ty)
            for (int tx = minx; tx < maxx; ++tx) {
                int64_t tile = (int64_t)ty * tiles_x + tx;
                isect_ids[cur] = (tile << 32) | (int64_t)(uint32_t)dbits;
                gaussian_ids[cur] = (int32_t)i;
                ++cur;
            }
    }
}

/* stable LSD radix sort of (key,id) pairs by key; ties keep emission (= ascending id) order */
void orc_sort_intersects(int64_t M, const int64_t *keys_in, const int32_t *ids_in, int64_t *keys_out, int32_t *ids_out)
{
    if (M == 0) return;
    uint64_t *ka = (uint64_t *)malloc(M * 8), *kb = (uint64_t *)malloc(M * 8);
    int32_t *ia = (int32_t *)malloc(M * 4), *ib = (int32_t *)malloc(M * 4);
    memcpy(ka, keys_in, M * 8); memcpy(ia, ids_in, M * 4);
    for (int pass = 0; pass < 8; ++pass) {
        int64_t cnt[257]; memset(cnt, 0, sizeof(cnt));
        int sh = pass * 8;
        for (int64_t i = 0; i < M; ++i) cnt[((ka[i] >> sh) & 255) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < M; ++i) { int64_t d = cnt[(ka[i] >> sh) & 255]++; kb[d] = ka[i]; ib[d] = ia[i]; }
        uint64_t *tk = ka; ka = kb; kb = tk; int32_t *ti = ia; ia = ib; ib = ti;
    }
    memcpy(keys_out, ka, M * 8); memcpy(ids_out, ia, M * 4);
    free(ka); free(kb); free(ia); free(ib);
}

void orc_get_tile_bin_edges(int64_t M, const int64_t *keys_sorted, int num_tiles, int32_t *tile_bins /* [T,2] */)
{
    memset(tile_bins, 0, (size_t)num_tiles * 2 * sizeof(int32_t));
    for (int64_t i = 0; i < M; ++i) {
        int32_t t = (int32_t)(keys_sorted[i] >> 32);
        if (i == 0) tile_bins[2 * t] = 0;
        else {
            int32_t tp = (int32_t)(keys_sorted[i - 1] >> 32);
            if (tp != t) { tile_bins[2 * tp + 1] = (int32_t)i; tile_bins[2 * t] = (int32_t)i; }
        }
        if (i == M - 1) tile_bins[2 * t + 1] = (int32_t)M;
    }
}

/* ---------------------------------------------------------------- A.4 rasterize forward --- */
/* colors[N,3]; extra[N] optional 4th channel (camera depth) composited with the same weights and
 * zero background (restates the reference's second rasterize pass, gc_model.py:191-202). */
void orc_rasterize_fwd(int H, int W, int tiles_x, int tiles_y,
                       const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                       const float *xys, const float *conics, const float *colors, const float *opacities,
                       const float *extra, const float *background,
                       float *out_img, float *out_extra, float *final_Ts, int32_t *final_index)
{
    ORC_PAR_ROWS
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            int tile = (i / TILE) * tiles_x + (j / TILE);
            (void)tiles_y;
            int start = tile_bins[2 * tile], end = tile_bins[2 * tile + 1];
            float px = (float)j, py = (float)i;
            float T = 1.f, r = 0.f, g = 0.f, b = 0.f, e = 0.f;
            int last = 0;
            for (int k = start; k < end; ++k) {
                int gid = gaussian_ids_sorted[k];
                float dx = xys[2 * gid] - px, dy = xys[2 * gid + 1] - py;
                const float *cn = conics + 3 * gid;
                float sigma = 0.5f * (cn[0] * dx * dx + cn[2] * dy * dy) + cn[1] * dx * dy;
                float alpha = fminf(ALPHA_CAP, opacities[gid] * expf(-sigma));
                if (sigma < 0.f || alpha < ALPHA_MIN) continue;
                float next_T = T * (1.f - alpha);
                if (next_T <= T_STOP) break;
                float vis = alpha * T;
                r += colors[3 * gid] * vis; g += colors[3 * gid + 1] * vis; b += colors[3 * gid + 2] * vis;
                if (extra) e += extra[gid] * vis;
                T = next_T;
                last = k;
            }
            int pix = i * W + j;
            final_Ts[pix] = T; final_index[pix] = last;
            out_img[3 * pix] = r + T * background[0];
            out_img[3 * pix + 1] = g + T * background[1];
            out_img[3 * pix + 2] = b + T * background[2];
            if (out_extra) out_extra[pix] = e;
        }
}

/* ---------------------------------------------------------------- A.5 rasterize backward -- */
/* v_out[H,W,3], v_out_alpha[H,W] (may be NULL) -> v_xy[N,2], v_conic[N,3], v_colors[N,3], v_opacity[N] */
void orc_rasterize_bwd(int H, int W, int tiles_x, int64_t N,
                       const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                       const float *xys, const float *conics, const float *colors, const float *opacities,
                       const float *background, const float *final_Ts, const int32_t *final_index,
                       const float *v_out, const float *v_out_alpha,
                       float *v_xy, float *v_conic, float *v_colors, float *v_opacity)
{
    memset(v_xy, 0, N * 2 * sizeof(float)); memset(v_conic, 0, N * 3 * sizeof(float));
    memset(v_colors, 0, N * 3 * sizeof(float)); memset(v_opacity, 0, N * sizeof(float));
    ORC_PAR_ROWS
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            int tile = (i / TILE) * tiles_x + (j / TILE);
            int start = tile_bins[2 * tile], end = tile_bins[2 * tile + 1];
            if (end <= start) continue;
            int pix = i * W + j;
            float px = (float)j, py = (float)i;
            float T_final = final_Ts[pix], T = T_final;
            int bin_final = final_index[pix];
            float vo0 = v_out[3 * pix], vo1 = v_out[3 * pix + 1], vo2 = v_out[3 * pix + 2];
            float voa = v_out_alpha ? v_out_alpha[pix] : 0.f;
            float bgdot = background[0] * vo0 + background[1] * vo1 + background[2] * vo2;
            float S0 = 0.f, S1 = 0.f, S2 = 0.f;
            for (int k = bin_final; k >= start; --k) {
                int gid = gaussian_ids_sorted[k];
                float dx = xys[2 * gid] - px, dy = xys[2 * gid + 1] - py;
                const float *cn = conics + 3 * gid;
                float sigma = 0.5f * (cn[0] * dx * dx + cn[2] * dy * dy) + cn[1] * dx * dy;
                float vis = expf(-sigma);
                float opac = opacities[gid];
                float araw = opac * vis;
                float alpha = fminf(ALPHA_CAP, araw);
                if (sigma < 0.f || alpha < ALPHA_MIN) continue;
                float ra = 1.f / (1.f - alpha);
                T *= ra;                               /* T before this splat */
                float fac = alpha * T;
                ORC_ADD(v_colors[3 * gid], fac * vo0); ORC_ADD(v_colors[3 * gid + 1], fac * vo1); ORC_ADD(v_colors[3 * gid + 2], fac * vo2);
                const float *c = colors + 3 * gid;
                float v_alpha = (c[0] * T - S0 * ra) * vo0 + (c[1] * T - S1 * ra) * vo1 + (c[2] * T - S2 * ra) * vo2;
                v_alpha += T_final * ra * voa;         /* alpha = 1 - T_final */
                v_alpha += -T_final * ra * bgdot;      /* background term */
                S0 += c[0] * fac; S1 += c[1] * fac; S2 += c[2] * fac;
                if (araw > ALPHA_CAP) continue;        /* capped: d alpha / d(opac,sigma) = 0 */
                float v_sigma = -opac * vis * v_alpha;
                ORC_ADD(v_conic[3 * gid], 0.5f * v_sigma * dx * dx);
                ORC_ADD(v_conic[3 * gid + 1], v_sigma * dx * dy);
                ORC_ADD(v_conic[3 * gid + 2], 0.5f * v_sigma * dy * dy);
                ORC_ADD(v_xy[2 * gid], v_sigma * (cn[0] * dx + cn[1] * dy));
                ORC_ADD(v_xy[2 * gid + 1], v_sigma * (cn[1] * dx + cn[2] * dy));
                ORC_ADD(v_opacity[gid], vis * v_alpha);
            }
        }
}
