// Layout and resampling kernels around the GridNet convolutions (all HBM-bound, one pass each).
//   vlg_nchw_to_padded / vlg_padded_to_nchw   (b,C,H,W) <-> halo-padded channels-last (conv.hip header)
//   vlg_fill_coords                           AddCoords channels          reference src/models/modules.py:65-96
//   vlg_upsample2x_fwd / _bwd                 nn.Upsample(x2, bilinear, align_corners=True)  modules.py:50
//   vlg_sum_partials                          per-block partial sums -> one scalar (PReLU slope gradients)
#include "common.h"

#define GO_BLOCK 256

__global__ void scale_scalar_kernel(float* v, float s) { v[0] *= s; }
__global__ void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ dst, int accumulate);

static unsigned go_blocks(int64_t n) {
    int64_t b = (n + GO_BLOCK - 1) / GO_BLOCK;
    return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// thread per (interior pixel, channel): reads NCHW (coalesced over x for fixed c), writes channels-last
__global__ __launch_bounds__(GO_BLOCK) void nchw_to_padded_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                  int b, int C, int H, int W, int cp) {
    const int64_t total = (int64_t)b * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GO_BLOCK) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C);
        const int64_t n = i / ((int64_t)W * H * C);
        dst[((n * (H + 2) + y + 1) * (W + 2) + x + 1) * cp + c] = src[i];
    }
}

__global__ __launch_bounds__(GO_BLOCK) void padded_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                  int b, int C, int H, int W, int cp) {
    const int64_t total = (int64_t)b * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GO_BLOCK) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C);
        const int64_t n = i / ((int64_t)W * H * C);
        dst[i] = src[((n * (H + 2) + y + 1) * (W + 2) + x + 1) * cp + c];
    }
}

// AddCoords: channel c0 varies along H, channel c0+1 along W, values k/(dim-1)*2 - 1
// (reference modules.py:69-70 hard-codes 256; SURVEY.md section 8 a4 verified the orientation numerically)
__global__ __launch_bounds__(GO_BLOCK) void fill_coords_kernel(float* __restrict__ dst, int b, int H, int W, int cp, int c0) {
    const int64_t total = (int64_t)b * H * W;
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GO_BLOCK) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int64_t n = i / ((int64_t)W * H);
        float* px = dst + ((n * (H + 2) + y + 1) * (W + 2) + x + 1) * cp;
        px[c0] = (float)y / (float)(H - 1) * 2.0f - 1.0f;
        px[c0 + 1] = (float)x / (float)(W - 1) * 2.0f - 1.0f;
    }
}

// align_corners=True source coordinate: src = dst * (in-1)/(out-1)  (ATen area_pixel_compute_scale)
__device__ __forceinline__ void up_src(int o, int in, float scale, int& i0, int& i1, float& l1) {
    const float s = scale * (float)o;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// thread per (output pixel, float4 of channels)
__global__ __launch_bounds__(GO_BLOCK) void upsample2x_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                  int b, int h, int w, int cp) {
    const int H = 2 * h, W = 2 * w, c4n = cp >> 2;
    const float sh = (float)(h - 1) / (float)(H - 1), sw = (float)(w - 1) / (float)(W - 1);
    const int64_t total = (int64_t)b * H * W * c4n;
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GO_BLOCK) {
        const int c = (int)(i % c4n) << 2;
        const int X = (int)((i / c4n) % W);
        const int Y = (int)((i / ((int64_t)c4n * W)) % H);
        const int64_t n = i / ((int64_t)c4n * W * H);
        int y0, y1, x0, x1; float ly, lx;
        up_src(Y, h, sh, y0, y1, ly);
        up_src(X, w, sw, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* base = in + (n * (h + 2)) * (int64_t)(w + 2) * cp + c;
        const float4 a = ld4(base + ((int64_t)(y0 + 1) * (w + 2) + x0 + 1) * cp), bq = ld4(base + ((int64_t)(y0 + 1) * (w + 2) + x1 + 1) * cp);
        const float4 cq = ld4(base + ((int64_t)(y1 + 1) * (w + 2) + x0 + 1) * cp), d = ld4(base + ((int64_t)(y1 + 1) * (w + 2) + x1 + 1) * cp);
        float4 o;
        o.x = hy * (hx * a.x + lx * bq.x) + ly * (hx * cq.x + lx * d.x);
        o.y = hy * (hx * a.y + lx * bq.y) + ly * (hx * cq.y + lx * d.y);
        o.z = hy * (hx * a.z + lx * bq.z) + ly * (hx * cq.z + lx * d.z);
        o.w = hy * (hx * a.w + lx * bq.w) + ly * (hx * cq.w + lx * d.w);
        st4(out + ((n * (H + 2) + Y + 1) * (int64_t)(W + 2) + X + 1) * cp + c, o);
    }
}

// gather form of the transpose: every input pixel collects the output pixels whose 2x2 stencil touches it
// (at most 3 output rows x 3 output columns around 2y, 2x), recomputing their weights - no atomics.
__global__ __launch_bounds__(GO_BLOCK) void upsample2x_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                                  int b, int h, int w, int cp, int accumulate) {
    const int H = 2 * h, W = 2 * w, c4n = cp >> 2;
    const float sh = (float)(h - 1) / (float)(H - 1), sw = (float)(w - 1) / (float)(W - 1);
    const int64_t total = (int64_t)b * h * w * c4n;
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GO_BLOCK) {
        const int c = (int)(i % c4n) << 2;
        const int x = (int)((i / c4n) % w);
        const int y = (int)((i / ((int64_t)c4n * w)) % h);
        const int64_t n = i / ((int64_t)c4n * w * h);
        float4 acc = f4_zero();
        const float* base = dout + (n * (H + 2)) * (int64_t)(W + 2) * cp + c;
        for (int Y = 2 * y - 2; Y <= 2 * y + 3; ++Y) {
            if (Y < 0 || Y >= H) continue;
            int y0, y1; float ly;
            up_src(Y, h, sh, y0, y1, ly);
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int X = 2 * x - 2; X <= 2 * x + 3; ++X) {
                if (X < 0 || X >= W) continue;
                int x0, x1; float lx;
                up_src(X, w, sw, x0, x1, lx);
                const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
                if (wx == 0.f) continue;
                const float4 g = ld4(base + ((int64_t)(Y + 1) * (W + 2) + X + 1) * cp);
                const float wgt = wy * wx;
                acc.x += wgt * g.x; acc.y += wgt * g.y; acc.z += wgt * g.z; acc.w += wgt * g.w;
            }
        }
        float* dst = din + ((n * (h + 2) + y + 1) * (int64_t)(w + 2) + x + 1) * cp + c;
        if (accumulate) acc = f4_add(acc, ld4(dst));
        st4(dst, acc);
    }
}

__global__ __launch_bounds__(GO_BLOCK) void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ dst,
                                                                int accumulate) {
    __shared__ float red[GO_BLOCK / 64];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += GO_BLOCK) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) dst[0] = accumulate ? dst[0] + s : s;
}

__global__ __launch_bounds__(GO_BLOCK) void add_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n4,
                                                           int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * GO_BLOCK) {
        float4 v = ld4(src + 4 * i);
        if (accumulate) v = f4_add(v, ld4(dst + 4 * i));
        st4(dst + 4 * i, v);
    }
}

// ---------------------------------------------------------------------------- HED helpers
// (frozen edge detector, reference src/models/hned.py:9-105; forward only)
// MaxPool2d(2,2) on padded NHWC, hned.py:20,28,38,48.  ReLU commutes with max, so the pool runs on the
// pre-activation tensor and the consumer applies the ReLU on load.
__global__ __launch_bounds__(GO_BLOCK) void maxpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out, int b, int h,
                                                             int w, int cp) {
    const int H = 2 * h, W = 2 * w, c4n = cp >> 2;
    const int64_t total = (int64_t)b * h * w * c4n;
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GO_BLOCK) {
        const int c = (int)(i % c4n) << 2;
        const int x = (int)((i / c4n) % w);
        const int y = (int)((i / ((int64_t)c4n * w)) % h);
        const int64_t n = i / ((int64_t)c4n * w * h);
        const float* p = in + ((n * (H + 2) + 2 * y + 1) * (int64_t)(W + 2) + 2 * x + 1) * cp + c;
        const float4 a = ld4(p), bq = ld4(p + cp), cq = ld4(p + (int64_t)(W + 2) * cp), d = ld4(p + (int64_t)(W + 3) * cp);
        float4 m;
        m.x = fmaxf(fmaxf(a.x, bq.x), fmaxf(cq.x, d.x)); m.y = fmaxf(fmaxf(a.y, bq.y), fmaxf(cq.y, d.y));
        m.z = fmaxf(fmaxf(a.z, bq.z), fmaxf(cq.z, d.z)); m.w = fmaxf(fmaxf(a.w, bq.w), fmaxf(cq.w, d.w));
        st4(out + ((n * (h + 2) + y + 1) * (int64_t)(w + 2) + x + 1) * cp + c, m);
    }
}

// 1x1 score convolution C -> 1 on relu(x) (hned.py:60-64, 90-94): one wavefront per pixel, shuffle reduction
__global__ __launch_bounds__(GO_BLOCK) void score1x1_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int b,
                                                           int H, int W, int C, int cp) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * GO_BLOCK + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * GO_BLOCK) >> 6;
    const int64_t total = (int64_t)b * H * W;
    for (int64_t i = wave; i < total; i += nw) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int64_t n = i / ((int64_t)W * H);
        const float* p = in + ((n * (H + 2) + y + 1) * (int64_t)(W + 2) + x + 1) * cp;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += fmaxf(p[c], 0.f) * w[c];
        s = wave_sum(s);
        if (lane == 0) out[i] = s + bias[0];
    }
}

// F.interpolate(size=(H,W), mode='bilinear', align_corners=False) source index (ATen area_pixel_compute_source_index)
__device__ __forceinline__ float hed_sample(const float* __restrict__ m, int h, int w, int Y, int X, float sy, float sx) {
    float fy = sy * ((float)Y + 0.5f) - 0.5f, fx = sx * ((float)X + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    return (1.f - ly) * ((1.f - lx) * m[y0 * w + x0] + lx * m[y0 * w + x1]) + ly * ((1.f - lx) * m[y1 * w + x0] + lx * m[y1 * w + x1]);
}

// five score maps (levels 0..4 at H>>k) -> d1..d5 = sigmoid(upsampled score), fuse = sigmoid(1x1 conv over the five)
// (hned.py:96-103); out is (6, b, H, W)
__global__ __launch_bounds__(GO_BLOCK) void hed_head_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                           const float* __restrict__ s2, const float* __restrict__ s3,
                                                           const float* __restrict__ s4, const float* __restrict__ cw,
                                                           const float* __restrict__ cb, float* __restrict__ out, int b, int H,
                                                           int W) {
    const float* maps[5] = {s0, s1, s2, s3, s4};
    const int64_t hw = (int64_t)H * W, total = (int64_t)b * hw;
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GO_BLOCK) {
        const int X = (int)(i % W), Y = (int)((i / W) % H);
        const int64_t n = i / hw;
        float f = cb[0];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int h = H >> k, w = W >> k;
            const float v = k == 0 ? maps[0][i] : hed_sample(maps[k] + n * h * w, h, w, Y, X, (float)h / (float)H, (float)w / (float)W);
            out[(int64_t)k * total + i] = 1.f / (1.f + expf(-v));
            f += cw[k] * v;
        }
        out[5 * total + i] = 1.f / (1.f + expf(-f));
    }
}

// backward of the 2x2 max-pool: the gradient of a pooled pixel goes to the first maximum of its window in
// row-major order (torch's tie rule); thread per (input pixel, float4 of channels), gather form
__global__ __launch_bounds__(GO_BLOCK) void maxpool2x2_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                                 float* __restrict__ din, int b, int h, int w, int cp) {
    const int H = 2 * h, W = 2 * w, c4n = cp >> 2;
    const int64_t total = (int64_t)b * h * w * c4n;
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GO_BLOCK) {
        const int c = (int)(i % c4n) << 2;
        const int x = (int)((i / c4n) % w);
        const int y = (int)((i / ((int64_t)c4n * w)) % h);
        const int64_t n = i / ((int64_t)c4n * w * h);
        const int64_t base = ((n * (H + 2) + 2 * y + 1) * (int64_t)(W + 2) + 2 * x + 1) * cp + c;
        const int64_t offs[4] = {0, cp, (int64_t)(W + 2) * cp, (int64_t)(W + 3) * cp};
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ld4(in + base + offs[k]);
        const float4 g = ld4(dout + ((n * (h + 2) + y + 1) * (int64_t)(w + 2) + x + 1) * cp + c);
        const float* gp = &g.x;
        float4 o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = f4_zero();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int best = 0;
            float bv = (&v[0].x)[e];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float t = (&v[k].x)[e];
                if (t > bv) { bv = t; best = k; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) (&o[k].x)[e] = (k == best) ? gp[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) st4(din + base + offs[k], o[k]);
    }
}

// mean |relu(a) - relu(b)| over C channels of the interior pixels of two padded tensors, and its gradient with respect
// to the PRE-activation a (VggLoss.forward, reference src/loss.py:43-49: features end in relu4_4)
__global__ __launch_bounds__(GO_BLOCK) void l1_relu_padded_kernel(const float* __restrict__ a, const float* __restrict__ bq,
                                                                 float* __restrict__ da, float* __restrict__ part, int64_t n4,
                                                                 float gscale) {
    __shared__ float red[GO_BLOCK / 64];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * GO_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * GO_BLOCK) {
        const float4 x = ld4(a + 4 * i), y = ld4(bq + 4 * i);
        const float* xp = &x.x; const float* yp = &y.x;
        float4 g;
        float* gp = &g.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = fmaxf(xp[e], 0.f) - fmaxf(yp[e], 0.f);
            acc += fabsf(d);
            gp[e] = xp[e] > 0.f ? (d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f)) : 0.f;
        }
        if (da != nullptr) st4(da + 4 * i, g);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

extern "C" int vlg_maxpool2x2_bwd(const float* in, const float* dout, float* din, int b, int h, int w, int cp, void* stream) {
    if (b < 1 || h < 1 || w < 1 || (cp & 3)) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(in) || !vlg_aligned16(dout) || !vlg_aligned16(din)) return VLG_ERR_ALIGN;
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(go_blocks((int64_t)b * h * w * (cp / 4))), dim3(GO_BLOCK), 0, (hipStream_t)stream, in, dout, din, b, h, w, cp);
    return vlg_last_error();
}

extern "C" int vlg_l1_relu_padded(const float* a, const float* b_, float* da, float* loss, float* scratch, int64_t rows, int cp,
                                  int64_t count, float grad_scale, void* stream) {
    // scratch >= 4096 floats; `count` = number of real elements (b*C*H*W) the mean is taken over
    if (rows < 1 || cp < 4 || (cp & 3) || count < 1) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(a) || !vlg_aligned16(b_) || (da && !vlg_aligned16(da))) return VLG_ERR_ALIGN;
    const int64_t n4 = rows * cp / 4;
    const unsigned blocks = go_blocks(n4);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(l1_relu_padded_kernel, dim3(blocks), dim3(GO_BLOCK), 0, s, a, b_, da, scratch, n4, grad_scale / (float)count);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(GO_BLOCK), 0, s, scratch, (int)blocks, loss, 0);
    hipLaunchKernelGGL(scale_scalar_kernel, dim3(1), dim3(1), 0, s, loss, 1.0f / (float)count);
    return vlg_last_error();
}

extern "C" int vlg_maxpool2x2(const float* in, float* out, int b, int h, int w, int cp, void* stream) {
    if (b < 1 || h < 1 || w < 1 || (cp & 3)) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(in) || !vlg_aligned16(out)) return VLG_ERR_ALIGN;
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(go_blocks((int64_t)b * h * w * (cp / 4))), dim3(GO_BLOCK), 0, (hipStream_t)stream, in, out, b, h, w, cp);
    return vlg_last_error();
}

extern "C" int vlg_score1x1_relu(const float* in, const float* w, const float* bias, float* out, int b, int H, int W, int C,
                                 int cp, void* stream) {
    if (b < 1 || H < 1 || W < 1 || C < 1 || C > cp) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL(score1x1_kernel, dim3(go_blocks((int64_t)b * H * W * 64)), dim3(GO_BLOCK), 0, (hipStream_t)stream, in, w, bias, out, b, H, W, C, cp);
    return vlg_last_error();
}

extern "C" int vlg_hed_head(const float* s0, const float* s1, const float* s2, const float* s3, const float* s4,
                            const float* combine_w, const float* combine_b, float* out, int b, int H, int W, void* stream) {
    if (b < 1 || H < 16 || W < 16 || (H & 15) || (W & 15)) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL(hed_head_kernel, dim3(go_blocks((int64_t)b * H * W)), dim3(GO_BLOCK), 0, (hipStream_t)stream, s0, s1, s2, s3, s4, combine_w, combine_b, out, b, H, W);
    return vlg_last_error();
}

extern "C" int vlg_add_rows(float* dst, const float* src, int64_t n, int accumulate, void* stream) {
    if (n < 4 || (n & 3)) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(dst) || !vlg_aligned16(src)) return VLG_ERR_ALIGN;
    hipLaunchKernelGGL(add_rows_kernel, dim3(go_blocks(n / 4)), dim3(GO_BLOCK), 0, (hipStream_t)stream, dst, src, n / 4, accumulate);
    return vlg_last_error();
}

extern "C" int vlg_nchw_to_padded(const float* src, float* dst, int b, int C, int H, int W, int cp, int coord_c0, void* stream) {
    if (b < 1 || C < 1 || H < 2 || W < 2 || cp < C || (cp & 3)) return VLG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nchw_to_padded_kernel, dim3(go_blocks((int64_t)b * C * H * W)), dim3(GO_BLOCK), 0, s, src, dst, b, C, H, W, cp);
    if (coord_c0 >= 0) {
        if (coord_c0 + 2 > cp) return VLG_ERR_SHAPE;
        hipLaunchKernelGGL(fill_coords_kernel, dim3(go_blocks((int64_t)b * H * W)), dim3(GO_BLOCK), 0, s, dst, b, H, W, cp, coord_c0);
    }
    return vlg_last_error();
}

extern "C" int vlg_padded_to_nchw(const float* src, float* dst, int b, int C, int H, int W, int cp, void* stream) {
    if (b < 1 || C < 1 || H < 1 || W < 1 || cp < C) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL(padded_to_nchw_kernel, dim3(go_blocks((int64_t)b * C * H * W)), dim3(GO_BLOCK), 0, (hipStream_t)stream, src, dst, b, C, H, W, cp);
    return vlg_last_error();
}

extern "C" int vlg_fill_coords(float* dst, int b, int H, int W, int cp, int coord_c0, void* stream) {
    if (b < 1 || H < 2 || W < 2 || coord_c0 < 0 || coord_c0 + 2 > cp) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL(fill_coords_kernel, dim3(go_blocks((int64_t)b * H * W)), dim3(GO_BLOCK), 0, (hipStream_t)stream, dst, b, H, W, cp, coord_c0);
    return vlg_last_error();
}

extern "C" int vlg_upsample2x_fwd(const float* in, float* out, int b, int h, int w, int cp, void* stream) {
    if (b < 1 || h < 2 || w < 2 || (cp & 3)) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(in) || !vlg_aligned16(out)) return VLG_ERR_ALIGN;
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(go_blocks((int64_t)b * 4 * h * w * (cp / 4))), dim3(GO_BLOCK), 0, (hipStream_t)stream, in, out, b, h, w, cp);
    return vlg_last_error();
}

extern "C" int vlg_upsample2x_bwd(const float* dout, float* din, int b, int h, int w, int cp, int accumulate, void* stream) {
    if (b < 1 || h < 2 || w < 2 || (cp & 3)) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(dout) || !vlg_aligned16(din)) return VLG_ERR_ALIGN;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(go_blocks((int64_t)b * h * w * (cp / 4))), dim3(GO_BLOCK), 0, (hipStream_t)stream, dout, din, b, h, w, cp, accumulate);
    return vlg_last_error();
}

extern "C" int vlg_sum_partials(const float* partials, int n, float* dst, int accumulate, void* stream) {
    if (n < 1) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(GO_BLOCK), 0, (hipStream_t)stream, partials, n, dst, accumulate);
    return vlg_last_error();
}
