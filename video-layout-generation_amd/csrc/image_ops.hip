// Pixel-space ops of the reference training step that exist verbatim in the reference and are
// therefore TRUE-parity targets (golden vectors in tests/golden come from the reference itself).
// All are HBM-bound single-pass stencils / streams; each fuses forward value and input gradient.
//
//   vlg_ce_nchw        nn.CrossEntropyLoss('mean') over (b,C,H,W)        reference src/trainer.py:124,250
//   vlg_l1_mean        nn.L1Loss()                                       reference src/trainer.py:130,248
//   vlg_gradient_loss  GradientLoss.forward                              reference src/loss.py:20-25
//   vlg_ssim_loss      SsimLoss.SSIM / forward                           reference src/loss.py:68-91
//   vlg_prep_input     normalise + 10-channel concat + shared flip       reference src/trainer.py:193-206
//
// Loss values leave through per-block partials in `scratch` and a tiny finalize kernel: no float
// atomics, bitwise reproducible.
#include "common.h"

#define IMG_BLOCK 256
#define IMG_MAX_BLOCKS 1024
#define IMG_MAX_TILES 65536
// scratch (floats): [0] = normaliser, [1] = spare, [4 ..) per-block partial sums (SSIM: one per tile)
#define IMG_SCRATCH (4 + IMG_MAX_TILES + IMG_MAX_BLOCKS)      // + per-block counts of scored targets (cross entropy)
#define IMG_COUNT_OFF (4 + IMG_MAX_TILES)

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(64) void img_finalize_kernel(const float* __restrict__ scratch, int nblocks,
                                                          float scale, float* __restrict__ loss) {
    float a = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) a += scratch[4 + i];
    a = wave_sum(a);
    if (threadIdx.x == 0) loss[0] = a * scale;
}

static unsigned img_blocks(int64_t n) {
    int64_t b = (n + IMG_BLOCK - 1) / IMG_BLOCK;
    if (b > IMG_MAX_BLOCKS) b = IMG_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ------------------------------------------------------------------ cross entropy, NCHW
// scored-target counts (target != ignore_index -100), one partial per block; every block of the main kernel sums the
// partials itself, so no single-block pass over the whole target map sits on the critical path
__global__ __launch_bounds__(IMG_BLOCK) void ce_count_kernel(const int64_t* __restrict__ target, int64_t n,
                                                             float* __restrict__ scratch) {
    __shared__ float red[IMG_BLOCK / 64];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * IMG_BLOCK)
        s += (target[i] != -100) ? 1.f : 0.f;
    s = block_sum(s, red);
    if (threadIdx.x == 0) scratch[IMG_COUNT_OFF + blockIdx.x] = s;       // counts are integers < 2^24: exact in fp32
}

// one thread per pixel; class c of pixel p sits at logits[(n*C + c)*hw + p] (coalesced per class)
__global__ __launch_bounds__(IMG_BLOCK) void ce_nchw_kernel(const float* __restrict__ logits,
                                                            const int64_t* __restrict__ target,
                                                            float* __restrict__ dlogits, float* __restrict__ scratch,
                                                            int64_t npix, int C, int64_t hw, float grad_scale,
                                                            int n_count_blocks) {
    __shared__ float red[IMG_BLOCK / 64];
    __shared__ float s_inv;
    {
        float c = 0.f;
        for (int i = threadIdx.x; i < n_count_blocks; i += IMG_BLOCK) c += scratch[IMG_COUNT_OFF + i];
        c = block_sum(c, red);
        if (threadIdx.x == 0) s_inv = 1.0f / fmaxf(c, 1.0f);
        __syncthreads();
    }
    const float inv_cnt = s_inv;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x; i < npix; i += (int64_t)gridDim.x * IMG_BLOCK) {
        const int64_t n = i / hw, p = i - n * hw;
        const float* row = logits + n * C * hw + p;
        const int64_t tg = target[i];
        const bool ignore = tg == -100;
        const int cls = (int)(tg < 0 ? 0 : (tg >= C ? C - 1 : tg));
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c * hw]);
        float se = 0.f, vt = 0.f;
        for (int c = 0; c < C; ++c) {
            const float v = row[c * hw];
            se += expf(v - mx);
            vt = (c == cls) ? v : vt;
        }
        if (!ignore) acc += logf(se) + mx - vt;
        if (dlogits != nullptr) {
            float* drow = dlogits + n * C * hw + p;
            const float gs = ignore ? 0.f : grad_scale * inv_cnt;
            for (int c = 0; c < C; ++c) {
                const float sm = expf(row[c * hw] - mx) / se;
                drow[c * hw] = gs * (sm - ((c == cls) ? 1.f : 0.f));
            }
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) scratch[4 + blockIdx.x] = acc * inv_cnt;
}

// ------------------------------------------------------------------ L1 mean
__global__ __launch_bounds__(IMG_BLOCK) void l1_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ da, float* __restrict__ scratch,
                                                       int64_t n, float gscale) {
    __shared__ float red[IMG_BLOCK / 64];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * IMG_BLOCK) {
        const float df = a[i] - b[i];
        acc += fabsf(df);
        if (da != nullptr) da[i] = gscale * sgn(df);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) scratch[4 + blockIdx.x] = acc;
}

// ------------------------------------------------------------------ GradientLoss
// e_v(h,w) = | |a[h+1,w]-a[h,w]| - |b[h+1,w]-b[h,w]| |, e_h likewise along w; loss = (sum e_v + sum e_h)/numel
__global__ __launch_bounds__(IMG_BLOCK) void gradient_loss_kernel(const float* __restrict__ a,
                                                                  const float* __restrict__ b,
                                                                  float* __restrict__ da, float* __restrict__ scratch,
                                                                  int64_t planes, int H, int W, float gscale) {
    __shared__ float red[IMG_BLOCK / 64];
    const int64_t hw = (int64_t)H * W, n = planes * hw;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * IMG_BLOCK) {
        const int64_t pl = i / hw;
        const int r = (int)(i - pl * hw);
        const int h = r / W, w = r - h * W;
        const float ac = a[i], bc = b[i];
        float g = 0.f;
        if (h + 1 < H) {   // pair (h, h+1): this pixel is the lower index, owns the loss term
            const float d_a = a[i + W] - ac, d_b = b[i + W] - bc;
            const float e = fabsf(d_a) - fabsf(d_b);
            acc += fabsf(e);
            g -= sgn(e) * sgn(d_a);
        }
        if (h > 0) {
            const float d_a = ac - a[i - W], d_b = bc - b[i - W];
            g += sgn(fabsf(d_a) - fabsf(d_b)) * sgn(d_a);
        }
        if (w + 1 < W) {
            const float d_a = a[i + 1] - ac, d_b = b[i + 1] - bc;
            const float e = fabsf(d_a) - fabsf(d_b);
            acc += fabsf(e);
            g -= sgn(e) * sgn(d_a);
        }
        if (w > 0) {
            const float d_a = ac - a[i - 1], d_b = bc - b[i - 1];
            g += sgn(fabsf(d_a) - fabsf(d_b)) * sgn(d_a);
        }
        if (da != nullptr) da[i] = gscale * g;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) scratch[4 + blockIdx.x] = acc;
}

// ------------------------------------------------------------------ SSIM (3x3 mean windows, no padding)
// A block owns a 32x32 pixel tile of one plane: x,y with a 2-pixel halo go to LDS, the 34x34
// window coefficients (dL/dmu_x, dL/dE[x^2], dL/dE[xy], pre-divided by 9) go to LDS, then every
// pixel sums the 3x3 windows that contain it.  Loss is counted once, by the tile owning the centre.
#define ST 32
__global__ __launch_bounds__(IMG_BLOCK) void ssim_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         float* __restrict__ dx, float* __restrict__ scratch,
                                                         int H, int W, int tiles_w, int tiles_per_plane,
                                                         float gscale) {
    __shared__ float xs[ST + 4][ST + 5], ys[ST + 4][ST + 5];
    __shared__ float ca[ST + 2][ST + 3], cb[ST + 2][ST + 3], cc[ST + 2][ST + 3];
    __shared__ float red[IMG_BLOCK / 64];
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const int64_t plane = blockIdx.x / tiles_per_plane;
    const int tl = blockIdx.x - (int)(plane * tiles_per_plane);
    const int h0 = (tl / tiles_w) * ST, w0 = (tl % tiles_w) * ST;
    const float* xp = x + plane * (int64_t)H * W;
    const float* yp = y + plane * (int64_t)H * W;
    for (int e = threadIdx.x; e < (ST + 4) * (ST + 4); e += IMG_BLOCK) {
        const int r = e / (ST + 4), c = e - r * (ST + 4);
        const int hh = h0 - 2 + r, ww = w0 - 2 + c;
        const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
        xs[r][c] = in ? xp[(int64_t)hh * W + ww] : 0.f;
        ys[r][c] = in ? yp[(int64_t)hh * W + ww] : 0.f;
    }
    __syncthreads();
    float acc = 0.f;
    for (int e = threadIdx.x; e < (ST + 2) * (ST + 2); e += IMG_BLOCK) {
        const int r = e / (ST + 2), c = e - r * (ST + 2);
        const int hc = h0 - 1 + r, wc = w0 - 1 + c;              // window centre
        float va = 0.f, vb = 0.f, vc = 0.f;
        if (hc >= 1 && hc <= H - 2 && wc >= 1 && wc <= W - 2) {
            float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) {
                    const float xv = xs[r + dr][c + dc], yv = ys[r + dr][c + dc];
                    sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
                }
            const float mux = sx / 9.f, muy = sy / 9.f;
            const float sgx = sxx / 9.f - mux * mux, sgy = syy / 9.f - muy * muy, sgxy = sxy / 9.f - mux * muy;
            const float A1 = 2.f * mux * muy + C1, A2 = 2.f * sgxy + C2;
            const float B1 = mux * mux + muy * muy + C1, B2 = sgx + sgy + C2;
            const float S = (A1 * A2) / (B1 * B2);
            const float val = (1.f - S) * 0.5f;
            const bool own = r >= 1 && r <= ST && c >= 1 && c <= ST;
            if (own) acc += fminf(fmaxf(val, 0.f), 1.f);
            if (val >= 0.f && val <= 1.f) {                        // clamp passes gradient inside [0,1]
                const float inv = 1.f / (B1 * B2);
                const float dS_dmu = (2.f * muy * (A2 - A1) - 2.f * mux * S * (B2 - B1)) * inv;
                const float dS_dexx = -S / B2;
                const float dS_dexy = 2.f * A1 * inv;
                const float dL = -0.5f * gscale;
                va = dL * dS_dmu / 9.f;
                vb = dL * dS_dexx * 2.f / 9.f;
                vc = dL * dS_dexy / 9.f;
            }
        }
        ca[r][c] = va; cb[r][c] = vb; cc[r][c] = vc;
    }
    __syncthreads();
    if (dx != nullptr) {
        for (int e = threadIdx.x; e < ST * ST; e += IMG_BLOCK) {
            const int r = e / ST, c = e - r * ST;
            const int hh = h0 + r, ww = w0 + c;
            if (hh < H && ww < W) {
                float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
                for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) { sa += ca[r + dr][c + dc]; sb += cb[r + dr][c + dc]; sc += cc[r + dr][c + dc]; }
                dx[plane * (int64_t)H * W + (int64_t)hh * W + ww] = sa + sb * xs[r + 2][c + 2] + sc * ys[r + 2][c + 2];
            }
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) scratch[4 + blockIdx.x] = acc;
}

// sums an arbitrary number of per-block partials (SSIM launches one block per tile)
__global__ __launch_bounds__(IMG_BLOCK) void img_finalize_big_kernel(const float* __restrict__ scratch, int64_t nblocks,
                                                                     float scale, float* __restrict__ loss) {
    __shared__ float red[IMG_BLOCK / 64];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < nblocks; i += IMG_BLOCK) a += scratch[4 + i];
    a = block_sum(a, red);
    if (threadIdx.x == 0) loss[0] = a * scale;
}

// ------------------------------------------------------------------ input preparation
// x10 = cat[e1, seg1, norm(frame1), norm(frame2), seg2, e2]; frame3 normalised; optional W flip of
// everything (the reference flips x, the frames on dim 3 and seg3 on its last dim: trainer.py:200-206)
__global__ __launch_bounds__(IMG_BLOCK) void prep_input_kernel(
    const float* __restrict__ e1, const float* __restrict__ seg1, const float* __restrict__ f1,
    const float* __restrict__ f2, const float* __restrict__ seg2, const float* __restrict__ e2,
    const float* __restrict__ f3, const int64_t* __restrict__ seg3, float* __restrict__ x10,
    float* __restrict__ f3o, int64_t* __restrict__ seg3o, int b, int H, int W, int flip) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int64_t hw = (int64_t)H * W, total = (int64_t)b * hw;
    for (int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * IMG_BLOCK) {
        const int64_t n = i / hw;
        const int r = (int)(i - n * hw);
        const int h = r / W, w = r - h * W;
        const int ws = flip ? W - 1 - w : w;                       // source column
        const int64_t s1 = n * hw + (int64_t)h * W + ws;           // index in 1-channel maps
        float* xo = x10 + n * 10 * hw + r;
        xo[0] = e1[s1];
        xo[hw] = seg1[s1];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t s3 = (n * 3 + c) * hw + (int64_t)h * W + ws;
            xo[(2 + c) * hw] = (f1[s3] - mean[c]) / stdv[c];
            xo[(5 + c) * hw] = (f2[s3] - mean[c]) / stdv[c];
            f3o[(n * 3 + c) * hw + r] = (f3[s3] - mean[c]) / stdv[c];
        }
        xo[8 * hw] = seg2[s1];
        xo[9 * hw] = e2[s1];
        seg3o[i] = seg3[s1];
    }
}

// per-channel affine on (b,C,H,W): dst = (src - shift[c]) * scale[c]   (C <= 4)
// forward:  img = (img - mean_arr) / std_arr          reference src/trainer.py:212  (scale = 1/std, shift = mean)
// backward: d img_raw = d img / std_arr               (shift = 0)
__global__ __launch_bounds__(IMG_BLOCK) void affine_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               int64_t n, int C, int64_t hw, float4 shift, float4 scale) {
    const float sh[4] = {shift.x, shift.y, shift.z, shift.w}, sc[4] = {scale.x, scale.y, scale.z, scale.w};
    for (int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * IMG_BLOCK) {
        const int c = (int)((i / hw) % C);
        dst[i] = (src[i] - sh[c]) * sc[c];
    }
}

extern "C" int vlg_affine_nchw(const float* src, float* dst, int b, int C, int64_t hw, const float* shift_host,
                               const float* scale_host, void* stream) {
    if (b < 1 || C < 1 || C > 4 || hw < 1) return VLG_ERR_SHAPE;
    float sh[4] = {0, 0, 0, 0}, sc[4] = {1, 1, 1, 1};
    for (int c = 0; c < C; ++c) { sh[c] = shift_host[c]; sc[c] = scale_host[c]; }
    const int64_t n = (int64_t)b * C * hw;
    hipLaunchKernelGGL(affine_nchw_kernel, dim3(img_blocks(n)), dim3(IMG_BLOCK), 0, (hipStream_t)stream, src, dst, n, C, hw,
                       make_float4(sh[0], sh[1], sh[2], sh[3]), make_float4(sc[0], sc[1], sc[2], sc[3]));
    return vlg_last_error();
}

extern "C" int vlg_image_loss_scratch(void) { return IMG_SCRATCH; }

extern "C" int vlg_ce_nchw(const float* logits, const int64_t* target, float* dlogits, float* loss, float* scratch,
                           int b, int C, int64_t hw, float grad_scale, void* stream) {
    if (b < 1 || C < 1 || hw < 1) return VLG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)b * hw;
    const unsigned blocks = img_blocks(npix);
    hipLaunchKernelGGL(ce_count_kernel, dim3(blocks), dim3(IMG_BLOCK), 0, s, target, npix, scratch);
    hipLaunchKernelGGL(ce_nchw_kernel, dim3(blocks), dim3(IMG_BLOCK), 0, s, logits, target, dlogits, scratch, npix, C,
                       hw, grad_scale, (int)blocks);
    hipLaunchKernelGGL(img_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, (int)blocks, 1.0f, loss);
    return vlg_last_error();
}

extern "C" int vlg_l1_mean(const float* a, const float* b, float* da, float* loss, float* scratch, int64_t n,
                           float grad_scale, void* stream) {
    if (n < 1) return VLG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = img_blocks(n);
    const float inv = 1.0f / (float)n;
    hipLaunchKernelGGL(l1_kernel, dim3(blocks), dim3(IMG_BLOCK), 0, s, a, b, da, scratch, n, grad_scale * inv);
    hipLaunchKernelGGL(img_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, (int)blocks, inv, loss);
    return vlg_last_error();
}

extern "C" int vlg_gradient_loss(const float* a, const float* b, float* da, float* loss, float* scratch, int planes,
                                 int H, int W, float grad_scale, void* stream) {
    if (planes < 1 || H < 1 || W < 1) return VLG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)planes * H * W;
    const unsigned blocks = img_blocks(n);
    const float inv = 1.0f / (float)n;
    hipLaunchKernelGGL(gradient_loss_kernel, dim3(blocks), dim3(IMG_BLOCK), 0, s, a, b, da, scratch, (int64_t)planes,
                       H, W, grad_scale * inv);
    hipLaunchKernelGGL(img_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, (int)blocks, inv, loss);
    return vlg_last_error();
}

extern "C" int vlg_ssim_loss(const float* x, const float* y, float* dx, float* loss, float* scratch, int b, int C,
                             int H, int W, float grad_scale, void* stream) {
    if (b < 1 || C < 1 || H < 3 || W < 3) return VLG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int tiles_w = (W + ST - 1) / ST, tiles_h = (H + ST - 1) / ST;
    const int64_t blocks = (int64_t)b * C * tiles_w * tiles_h;
    if (blocks > IMG_MAX_TILES) return VLG_ERR_SHAPE;      // one partial per tile must fit the scratch
    // mean over b*(H-2)*(W-2) windows per channel, summed over channels (loss.py:89-91)
    const float inv = 1.0f / ((float)b * (float)(H - 2) * (float)(W - 2));
    hipLaunchKernelGGL(ssim_kernel, dim3((unsigned)blocks), dim3(IMG_BLOCK), 0, s, x, y, dx, scratch, H, W, tiles_w,
                       tiles_w * tiles_h, grad_scale * inv);
    hipLaunchKernelGGL(img_finalize_big_kernel, dim3(1), dim3(IMG_BLOCK), 0, s, scratch, blocks, inv, loss);
    return vlg_last_error();
}

extern "C" int vlg_prep_input(const float* e1, const float* seg1, const float* frame1, const float* frame2,
                              const float* seg2, const float* e2, const float* frame3, const int64_t* seg3,
                              float* x10, float* frame3_out, int64_t* seg3_out, int b, int H, int W, int flip,
                              void* stream) {
    if (b < 1 || H < 1 || W < 1) return VLG_ERR_SHAPE;
    const int64_t total = (int64_t)b * H * W;
    hipLaunchKernelGGL(prep_input_kernel, dim3(img_blocks(total)), dim3(IMG_BLOCK), 0, (hipStream_t)stream, e1, seg1,
                       frame1, frame2, seg2, e2, frame3, seg3, x10, frame3_out, seg3_out, b, H, W, flip);
    return vlg_last_error();
}

// ---- autoregressive rollout (reference src/trainer.py:453-476)
// seg_next = torch.argmax(seg_next, dim=1).unsqueeze_(1).float()      trainer.py:467   (first maximum wins, as torch)
__global__ __launch_bounds__(IMG_BLOCK) void argmax_nchw_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                               int64_t n, int C, int64_t hw) {
    for (int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * IMG_BLOCK) {
        const int64_t b = i / hw, r = i - b * hw;
        const float* p = logits + b * C * hw + r;
        float best = p[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = p[c * hw];
            if (v > best) { best = v; arg = c; }
        }
        out[i] = (float)arg;
    }
}

extern "C" int vlg_argmax_nchw(const float* logits, float* out, int b, int C, int64_t hw, void* stream) {
    if (b < 1 || C < 1 || hw < 1) return VLG_ERR_SHAPE;
    const int64_t n = (int64_t)b * hw;
    hipLaunchKernelGGL(argmax_nchw_kernel, dim3(img_blocks(n)), dim3(IMG_BLOCK), 0, (hipStream_t)stream, logits, out, n, C, hw);
    return vlg_last_error();
}

// x = cat[e_a, seg_a, img_a, img_b, seg_b, e_b] of frames that are ALREADY normalised: the rollout's input
// (trainer.py:461 with the channel order of the training input, trainer.py:197 - Appendix A-10 repaired)
__global__ __launch_bounds__(IMG_BLOCK) void rollout_input_kernel(const float* __restrict__ ea, const float* __restrict__ sa,
                                                                 const float* __restrict__ ia, const float* __restrict__ ib,
                                                                 const float* __restrict__ sb, const float* __restrict__ eb,
                                                                 float* __restrict__ x10, int64_t n, int64_t hw) {
    for (int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * IMG_BLOCK) {
        const int64_t b = i / hw, r = i - b * hw;
        float* xo = x10 + b * 10 * hw + r;
        xo[0] = ea[i];
        xo[hw] = sa[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xo[(2 + c) * hw] = ia[(b * 3 + c) * hw + r];
            xo[(5 + c) * hw] = ib[(b * 3 + c) * hw + r];
        }
        xo[8 * hw] = sb[i];
        xo[9 * hw] = eb[i];
    }
}

extern "C" int vlg_rollout_input(const float* e_a, const float* seg_a, const float* img_a, const float* img_b,
                                 const float* seg_b, const float* e_b, float* x10, int b, int64_t hw, void* stream) {
    if (b < 1 || hw < 1) return VLG_ERR_SHAPE;
    const int64_t n = (int64_t)b * hw;
    hipLaunchKernelGGL(rollout_input_kernel, dim3(img_blocks(n)), dim3(IMG_BLOCK), 0, (hipStream_t)stream, e_a, seg_a, img_a,
                       img_b, seg_b, e_b, x10, n, hw);
    return vlg_last_error();
}
