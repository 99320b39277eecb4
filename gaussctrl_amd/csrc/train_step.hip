// train_step.hip -- the loss and optimiser of the 500-iteration splat optimisation that follows the edit
// (SURVEY.md 8a row A8 / 8f-1), as streaming HIP kernels for gfx950.
//
// Replaces, on the reference's training iteration (/root/reference/gaussctrl/gc_trainer.py:257-301):
//   * SplatfactoModel.get_loss_dict (inherited via gc_pipeline.py:284-285): (1-l) * L1 + l * (1 - SSIM), l = 0.2, SSIM with an
//     11x11 gaussian window (sigma 1.5), and its backward to the rendered image;
//   * the Adam steps of the six Gaussian parameter groups (gc_config.py:58-87: eps 1e-15, per-group learning rates) --
//     7 x 4 B x 59 = 1 652 B / Gaussian / step of HBM traffic, more than the render itself.
//
// Everything is HBM-bound: the SSIM windows are separable (2 x 11 taps) and run on LDS tiles; Adam is one fused
// read-modify-write of (param, grad, m, v) with 16-byte accesses and the bias correction folded into two scalars.
#include "common.h"

namespace {

constexpr int WIN = 11, HALF = 5;
constexpr int TS = 16;   // output tile
constexpr int NSLOT = 32;  // partial-sum slots of the two loss scalars ...
constexpr int SLOT_PITCH = 32;   // ... one 128-byte line each: atomics to one cache line serialise in its L2 channel whatever the address

struct Gauss { float w[WIN]; };

// 5 separable 11x11 gaussian blurs in one pass: mu_x, mu_y, E[x^2], E[y^2], E[xy] of one channel plane; then the per-pixel SSIM
// and its partial derivatives w.r.t. (mu_x, E[x^2], E[xy]).  valid = 1: the SSIM map exists only where the 11x11 window lies inside
// the image ((H-10) x (W-10) pixels: pytorch_msssim's unpadded convolution, what splatfacto's loss uses); valid = 0: zero-padded
// 'same' windows on all H x W pixels (conv2d(padding=5), the 3DGS convention).
__global__ __launch_bounds__(256) void k_ssim_stats(const float *__restrict__ x, const float *__restrict__ y, int H, int W, int C, int valid,
                                                    Gauss gw, float *__restrict__ ssim_sum, float *__restrict__ dmu,
                                                    float *__restrict__ dxx, float *__restrict__ dxy, float *__restrict__ l1_sum)
{
    // ssim_sum / l1_sum: NSLOT partial sums each (slot = workgroup index mod NSLOT).  3072 workgroups adding into ONE address cost
    // ~12 ns per atomic, i.e. most of this kernel's 85 us (round 3); spread over 32 addresses they cost ~1 us; k_ssim_grad folds them.
    __shared__ float sx[TS + 2 * HALF][TS + 2 * HALF + 1], sy[TS + 2 * HALF][TS + 2 * HALF + 1];
    __shared__ float hb[5][TS + 2 * HALF][TS + 1];
    __shared__ float red[2][4];
    const int img = blockIdx.z / C, c = blockIdx.z - img * C, tx0 = blockIdx.x * TS, ty0 = blockIdx.y * TS, tid = threadIdx.x;
    {   // image `img` of a batch (gc_l1_ssim_fwd_bwd_views): [B][H][W][C] tensors, per-image slot lines
        const size_t o = (size_t)img * H * W * C;
        x += o; y += o; dmu += o; dxx += o; dxy += o;
        ssim_sum += (size_t)img * 2 * NSLOT * SLOT_PITCH; l1_sum += (size_t)img * 2 * NSLOT * SLOT_PITCH;
    }
    constexpr int TW = TS + 2 * HALF;
    for (int i = tid; i < TW * TW; i += 256) {
        const int ly = i / TW, lx = i - ly * TW;
        const int gy = ty0 + ly - HALF, gx = tx0 + lx - HALF;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        sx[ly][lx] = in ? x[((size_t)gy * W + gx) * C + c] : 0.f;
        sy[ly][lx] = in ? y[((size_t)gy * W + gx) * C + c] : 0.f;
    }
    __syncthreads();
    // horizontal pass: TW rows x TS columns
    for (int i = tid; i < TW * TS; i += 256) {
        const int ly = i / TS, lx = i - ly * TS;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float vx = sx[ly][lx + k], vy = sy[ly][lx + k], wk = gw.w[k];
            a += wk * vx; b += wk * vy; aa += wk * vx * vx; bb += wk * vy * vy; ab += wk * vx * vy;
        }
        hb[0][ly][lx] = a; hb[1][ly][lx] = b; hb[2][ly][lx] = aa; hb[3][ly][lx] = bb; hb[4][ly][lx] = ab;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int gx = tx0 + lx, gy = ty0 + ly;
    float ssim = 0.f, l1 = 0.f;
    const bool inside = gx < W && gy < H;
    const bool has_win = inside && (!valid || (gx >= HALF && gx < W - HALF && gy >= HALF && gy < H - HALF));
    if (inside && !has_win) {       // no SSIM term for this pixel: zero partials (k_ssim_grad gathers them), L1 still counts
        const size_t o = ((size_t)gy * W + gx) * C + c;
        dmu[o] = 0.f; dxx[o] = 0.f; dxy[o] = 0.f;
        l1 = fabsf(sx[ly + HALF][lx + HALF] - sy[ly + HALF][lx + HALF]);
    }
    if (has_win) {
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float wk = gw.w[k];
            m1 += wk * hb[0][ly + k][lx]; m2 += wk * hb[1][ly + k][lx];
            e11 += wk * hb[2][ly + k][lx]; e22 += wk * hb[3][ly + k][lx]; e12 += wk * hb[4][ly + k][lx];
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float s11 = e11 - m1 * m1, s22 = e22 - m2 * m2, s12 = e12 - m1 * m2;
        const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * s12 + C2, B1 = m1 * m1 + m2 * m2 + C1, B2 = s11 + s22 + C2;
        const float inv = 1.f / (B1 * B2);
        ssim = A1 * A2 * inv;
        // partials w.r.t. the x-side statistics (y = target is constant): via m1, e11 (through s11), e12 (through s12)
        const float d_s12 = 2.f * A1 * inv;                 // dS/ds12
        const float d_s11 = -ssim / B2;                     // dS/ds11
        const float d_m1 = (2.f * m2 * A2 - 2.f * m1 * ssim * B2) * inv   // direct through A1, B1
                           + d_s11 * (-2.f * m1) + d_s12 * (-m2);          // through s11 = e11 - m1^2, s12 = e12 - m1 m2
        const size_t o = ((size_t)gy * W + gx) * C + c;
        dmu[o] = d_m1; dxx[o] = d_s11; dxy[o] = d_s12;
        l1 = fabsf(sx[ly + HALF][lx + HALF] - sy[ly + HALF][lx + HALF]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ssim += __shfl_xor(ssim, d, 64); l1 += __shfl_xor(l1, d, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = ssim; red[1][tid >> 6] = l1; }
    __syncthreads();
    if (tid == 0) {
        const int slot = (int)((blockIdx.x + blockIdx.y * 5 + c * 11) & (NSLOT - 1));
        unsafeAtomicAdd(ssim_sum + slot * SLOT_PITCH, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        unsafeAtomicAdd(l1_sum + slot * SLOT_PITCH, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// v_x = (1-l)/n * sign(x - y) - l/n * ( G*(dmu) + 2 x G*(dxx) + y G*(dxy) )    (G = the same separable window, zero padded)
__global__ __launch_bounds__(256) void k_ssim_grad(const float *__restrict__ x, const float *__restrict__ y, int H, int W, int C,
                                                   Gauss gw, const float *__restrict__ dmu, const float *__restrict__ dxx,
                                                   const float *__restrict__ dxy, float lambda_, float inv_n, float inv_n_ssim, float scale,
                                                   float *__restrict__ v_x, const float *__restrict__ slots, float *__restrict__ loss_out)
{
    const int img = blockIdx.z / C;
    {
        const size_t o = (size_t)img * H * W * C;
        x += o; y += o; dmu += o; dxx += o; dxy += o; v_x += o;
        slots += (size_t)img * 2 * NSLOT * SLOT_PITCH; loss_out += 2 * img;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == img * C && threadIdx.x < 64) {      // fold the partial sums of k_ssim_stats
        float v = slots[threadIdx.x * SLOT_PITCH];     // slots [0, NSLOT): SSIM, [NSLOT, 2 NSLOT): L1
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if ((threadIdx.x & 31) == 0) loss_out[threadIdx.x >> 5] = v;
    }
    constexpr int TW = TS + 2 * HALF;
    __shared__ float s[3][TW][TW + 1];
    __shared__ float hb[3][TW][TS + 1];
    const int c = blockIdx.z - img * C, tx0 = blockIdx.x * TS, ty0 = blockIdx.y * TS, tid = threadIdx.x;
    for (int i = tid; i < TW * TW; i += 256) {
        const int ly = i / TW, lx = i - ly * TW;
        const int gy = ty0 + ly - HALF, gx = tx0 + lx - HALF;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = ((size_t)gy * W + gx) * C + c;
        s[0][ly][lx] = in ? dmu[o] : 0.f; s[1][ly][lx] = in ? dxx[o] : 0.f; s[2][ly][lx] = in ? dxy[o] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < TW * TS; i += 256) {
        const int ly = i / TS, lx = i - ly * TS;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) { const float wk = gw.w[k]; a += wk * s[0][ly][lx + k]; b += wk * s[1][ly][lx + k]; d += wk * s[2][ly][lx + k]; }
        hb[0][ly][lx] = a; hb[1][ly][lx] = b; hb[2][ly][lx] = d;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int gx = tx0 + lx, gy = ty0 + ly;
    if (gx >= W || gy >= H) return;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < WIN; ++k) { const float wk = gw.w[k]; a += wk * hb[0][ly + k][lx]; b += wk * hb[1][ly + k][lx]; d += wk * hb[2][ly + k][lx]; }
    const size_t o = ((size_t)gy * W + gx) * C + c;
    const float xv = x[o], yv = y[o];
    const float dssim = a + 2.f * xv * b + yv * d;
    const float sgn = xv > yv ? 1.f : (xv < yv ? -1.f : 0.f);
    v_x[o] = scale * ((1.f - lambda_) * inv_n * sgn - lambda_ * inv_n_ssim * dssim);
}

// Adam (torch.optim.Adam semantics, no weight decay / amsgrad): one fused pass, 16 bytes per lane per stream.
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                              float *__restrict__ v, int64_t n4, int64_t n, float beta1, float beta2, float eps,
                                              float step_size, float inv_sqrt_bc2)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4 *>(p)[i], gg = reinterpret_cast<const float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
#define GC_ADAM1(c)                                                     \
        mm.c = beta1 * mm.c + (1.f - beta1) * gg.c;                     \
        vv.c = beta2 * vv.c + (1.f - beta2) * gg.c * gg.c;              \
        pp.c -= step_size * mm.c / (sqrtf(vv.c) * inv_sqrt_bc2 + eps);
        GC_ADAM1(x) GC_ADAM1(y) GC_ADAM1(z) GC_ADAM1(w)
        reinterpret_cast<float4 *>(p)[i] = pp; reinterpret_cast<float4 *>(m)[i] = mm; reinterpret_cast<float4 *>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail (n not a multiple of 4)
        const int64_t i = n4 * 4 + threadIdx.x;
        const float gi = g[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi, vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

Gauss make_window()
{
    Gauss gw;
    float s = 0.f;
    for (int k = 0; k < WIN; ++k) { const float d = (float)(k - HALF); gw.w[k] = expf(-(d * d) / (2.f * 1.5f * 1.5f)); s += gw.w[k]; }
    for (int k = 0; k < WIN; ++k) gw.w[k] /= s;
    return gw;
}

}  // namespace

extern "C" {

size_t gc_l1_ssim_workspace_bytes(int H, int W, int C) { return sizeof(float) * (3 * (size_t)H * W * C + 32 + 2 * NSLOT * SLOT_PITCH); }
size_t gc_l1_ssim_views_workspace_bytes(int B, int H, int W, int C)
{
    return sizeof(float) * ((size_t)(B > 0 ? B : 1) * (3 * (size_t)H * W * C + 2 * NSLOT * SLOT_PITCH) + 32);
}

static int l1_ssim_impl(const char *what, int B, const float *pred, const float *target, int H, int W, int C, float lambda_, float grad_scale,
                        int valid_window, float *loss_out, float *v_pred, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!(pred && target && loss_out && v_pred && workspace)) { gc::set_error("%s: null argument", what); return GC_EINVAL; }
    if (valid_window && !(H > 2 * HALF && W > 2 * HALF)) { gc::set_error("%s: valid-window SSIM needs an image larger than 11 x 11", what); return GC_EINVAL; }
    if ((int64_t)B * C > 65535) { gc::set_error("%s: at most 65535 image planes per call", what); return GC_EINVAL; }
    hipStream_t s = gc::S(stream);
    const size_t n = (size_t)H * W * C;
    // workspace: dmu / dxx / dxy as [B][n] each, then B slot blocks (each block's lines start on a 128-byte boundary of the float array)
    float *dmu = (float *)workspace, *dxx = dmu + n * B, *dxy = dxx + n * B, *slots = dxy + (n * B + 31) / 32 * 32;
    static_assert(2 * NSLOT == 64, "k_ssim_grad folds the slots with one wave");
    if (hipMemsetAsync(slots, 0, (size_t)B * 2 * NSLOT * SLOT_PITCH * sizeof(float), s) != hipSuccess) return GC_ELAUNCH;
    const Gauss gw = make_window();
    dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, C * B);
    const size_t n_ssim = valid_window ? (size_t)(H - 2 * HALF) * (W - 2 * HALF) * C : n;
    hipLaunchKernelGGL(k_ssim_stats, grid, dim3(256), 0, s, pred, target, H, W, C, valid_window, gw, slots, dmu, dxx, dxy, slots + NSLOT * SLOT_PITCH);
    hipLaunchKernelGGL(k_ssim_grad, grid, dim3(256), 0, s, pred, target, H, W, C, gw, dmu, dxx, dxy, lambda_, 1.f / (float)n,
                       1.f / (float)n_ssim, grad_scale, v_pred, slots, loss_out);
    return gc::check_launch(what);
}

/* loss = (1-lambda)*mean|x-y| + lambda*(1 - mean SSIM(x,y)); pred x / target y: float32 [H,W,C] channels-last.
 * valid_window = 1: SSIM averaged over the (H-10) x (W-10) x C pixels with a full window (pytorch_msssim, splatfacto's loss);
 * 0: zero-padded windows, H x W x C pixels.
 * loss_out: device float[2] = {sum SSIM map, sum |x-y|} (the caller forms the scalar: no host sync here);
 * v_pred: d(loss)/d(pred) * grad_scale, same layout.  workspace >= gc_l1_ssim_workspace_bytes. */
int gc_l1_ssim_fwd_bwd(const float *pred, const float *target, int H, int W, int C, float lambda_, float grad_scale, int valid_window,
                       float *loss_out, float *v_pred, void *workspace, size_t workspace_bytes, void *stream)
{
    if (workspace_bytes < gc_l1_ssim_workspace_bytes(H, W, C)) { gc::set_error("gc_l1_ssim_fwd_bwd: workspace too small"); return GC_ENOSPC; }
    return l1_ssim_impl("gc_l1_ssim_fwd_bwd", 1, pred, target, H, W, C, lambda_, grad_scale, valid_window, loss_out, v_pred, workspace,
                        workspace_bytes, stream);
}

/* B image pairs in one pair of launches: pred / target / v_pred [B][H][W][C], loss_out [B][2] (each view's own sums: every view's loss is
 * the mean over ITS pixels, as B calls of the single-image form would give); workspace >= gc_l1_ssim_views_workspace_bytes. */
int gc_l1_ssim_fwd_bwd_views(int B, const float *pred, const float *target, int H, int W, int C, float lambda_, float grad_scale,
                             int valid_window, float *loss_out, float *v_pred, void *workspace, size_t workspace_bytes, void *stream)
{
    GC_REQUIRE(B >= 1, "bad arguments");
    if (workspace_bytes < gc_l1_ssim_views_workspace_bytes(B, H, W, C)) { gc::set_error("gc_l1_ssim_fwd_bwd_views: workspace too small"); return GC_ENOSPC; }
    return l1_ssim_impl("gc_l1_ssim_fwd_bwd_views", B, pred, target, H, W, C, lambda_, grad_scale, valid_window, loss_out, v_pred, workspace,
                        workspace_bytes, stream);
}

/* torch.optim.Adam step (betas, eps as given; step = 1-based iteration count) on one flat fp32 tensor. */
int gc_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, void *stream)
{
    GC_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "bad arguments");
    if (n == 0) return GC_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const int64_t n4 = n / 4;
    const unsigned grid = (unsigned)std::min<int64_t>((n4 + 255) / 256 + 1, 256 * 8);
    hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, gc::S(stream), param, grad, exp_avg, exp_avg_sq, n4, n, beta1, beta2, eps,
                       step_size, inv_sqrt_bc2);
    return gc::check_launch("gc_adam_step");
}

}  // extern "C"
