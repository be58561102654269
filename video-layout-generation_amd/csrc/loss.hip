// Fused layout losses: softmax cross-entropy + smooth-L1 + IoU, forward AND backward in one
// pass over the head outputs (HBM-bound; 2 * 4*(C+4) bytes per token + 44 bytes of targets).
//
//   total = w_reg * mean_valid( smoothL1(sigmoid(raw) - tgt) / 4 )
//         + w_iou * mean_valid( 1 - IoU(sigmoid(raw), tgt) )
//         + w_ce  * mean_valid( CE(logits, tgt_class) )
//   CE    = nn.CrossEntropyLoss(reduction='mean') arithmetic  (reference src/trainer.py:124,250)
//   40/20/10 weighting                                        (reference src/trainer.py:248-251)
//   smooth-L1 / IoU on (cx,cy,w,h) boxes: SELF-ORACLE (no reference counterpart)
//
// A block of 256 threads owns 128 tokens at a time.  Their head outputs (one 96-B row each) are staged through LDS, so HBM
// sees full 16-B-per-lane coalesced loads / stores (round 1-2 had one lane walk one 96-B row: 64 lanes of a load instruction
// touched 48 cache lines); lane t < 128 then computes token t from its LDS row (row stride 28 floats: the 16 lanes of a
// ds_read_b128 phase hit 16 distinct 4-bank groups), so the class softmax needs no cross-lane traffic.  The four loss sums
// are reduced per wavefront with shuffles, per block through LDS, and leave as one partial per block.
// Every block counts the valid slots ITSELF (the mean's denominator scales every gradient, so it must be known before the
// first store; M floats = 128 KB from L2, the same exact integer sum in every block), with its row and target loads already
// in flight and 16 independent 16-B loads per thread outstanding (round 2's loop waited for one L2 round trip per 16 B: 17 of
// the kernel's 26 us at M = 32 768).  The LAST block to finish (an integer ticket in scratch[1], which it resets) folds the
// partials into loss_out[4] in block order - no float atomics, reproducible, and no second launch.
#include "common.h"

#define LOSS_BLOCK 256
#define LOSS_TOK 128                 /* tokens per block pass */
#define LOSS_ROWF 28                 /* LDS row stride, floats (NOUT = 24 + 4 pad) */
#define LOSS_MAX_BLOCKS 1024
#define NCLS 20
#define NOUT (NCLS + 4)
#define LOSS_F4 (NOUT / 4)           /* float4 per row */
#define LOSS_NV (LOSS_TOK * LOSS_F4 / LOSS_BLOCK)     /* staged float4 per thread */

// scratch layout (floats): [1] = ticket counter (int, zero between launches); [4 .. 4+4*LOSS_MAX_BLOCKS) = per-block partials
#define LOSS_SCRATCH (4 + 4 * LOSS_MAX_BLOCKS)

__global__ __launch_bounds__(LOSS_BLOCK) void layout_loss_kernel(
    const float* __restrict__ out, int ld, const int64_t* __restrict__ tgt_class,
    const float* __restrict__ tgt_box, const float* __restrict__ valid, float* __restrict__ dout,
    float* __restrict__ scratch, float* __restrict__ loss_out, int B, int T, int N, float beta, float iou_eps,
    float w_reg, float w_iou, float w_ce) {
    __shared__ __attribute__((aligned(16))) float rows[LOSS_TOK * LOSS_ROWF];
    __shared__ float red[LOSS_BLOCK / 64];
    __shared__ float cnt_sh;
    __shared__ int last_sh;
    const int64_t M = (int64_t)B * T * N;
    const int tid = threadIdx.x;
    float inv_cnt = 0.f;
    float s_reg = 0.f, s_iou = 0.f, s_ce = 0.f;
    for (int64_t m0 = (int64_t)blockIdx.x * LOSS_TOK; m0 < M; m0 += (int64_t)gridDim.x * LOSS_TOK) {
        // ---- this pass's rows and targets: requested first, consumed after the count
        float4 stage[LOSS_NV];
#pragma unroll
        for (int i = 0; i < LOSS_NV; ++i) {
            const int f = tid + i * LOSS_BLOCK, tok = f / LOSS_F4, c = f - tok * LOSS_F4;
            stage[i] = (m0 + tok < M) ? ld4(out + (m0 + tok) * ld + 4 * c) : f4_zero();
        }
        const int64_t m = m0 + tid;
        const bool mine = tid < LOSS_TOK && m < M;
        float w = 0.f;
        int64_t cls = 0;
        float4 tb = f4_zero();
        if (mine) {
            const int t = (int)(m % T);
            const int64_t bn = m / T;
            const int n = (int)(bn % N);
            const int64_t b = bn / N;
            const int64_t src = (b * T + t) * N + n;
            w = valid[src];
            cls = tgt_class[src];
            tb = ld4(tgt_box + src * 4);
        }
        if (m0 == (int64_t)blockIdx.x * LOSS_TOK) {
            // 1 / max(#valid, 1): sums of 0/1 flags are exact in fp32 in any order, so every block gets the same value
            float c = 0.f;
            const int64_t n4 = M >> 2;
            int64_t i = tid;
            for (; i + 15 * LOSS_BLOCK < n4; i += 16 * LOSS_BLOCK) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = ld4(valid + 4 * (i + u * LOSS_BLOCK));
#pragma unroll
                for (int u = 0; u < 16; ++u) c += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            }
            for (; i < n4; i += LOSS_BLOCK) {
                const float4 v = ld4(valid + 4 * i);
                c += (v.x + v.y) + (v.z + v.w);
            }
            for (int64_t j = 4 * n4 + tid; j < M; j += LOSS_BLOCK) c += valid[j];
            c = block_sum(c, red);
            if (tid == 0) cnt_sh = 1.0f / fmaxf(c, 1.0f);
            __syncthreads();
            inv_cnt = cnt_sh;
        }
#pragma unroll
        for (int i = 0; i < LOSS_NV; ++i) {
            const int f = tid + i * LOSS_BLOCK, tok = f / LOSS_F4, c = f - tok * LOSS_F4;
            st4(rows + tok * LOSS_ROWF + 4 * c, stage[i]);
        }
        __syncthreads();
        if (mine) {
            float* row = rows + tid * LOSS_ROWF;
            float v[NOUT];
#pragma unroll
            for (int c = 0; c < LOSS_F4; ++c) {
                const float4 q = ld4(row + 4 * c);
                v[4 * c] = q.x; v[4 * c + 1] = q.y; v[4 * c + 2] = q.z; v[4 * c + 3] = q.w;
            }
            float g[NOUT];
            // ---- cross entropy over the first NCLS outputs
            cls = cls < 0 ? 0 : (cls >= NCLS ? NCLS - 1 : cls);
            float mx = v[0];
#pragma unroll
            for (int c = 1; c < NCLS; ++c) mx = fmaxf(mx, v[c]);
            float se = 0.f, vt = 0.f;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                g[c] = expf(v[c] - mx);
                se += g[c];
                vt = (c == cls) ? v[c] : vt;
            }
            const float ce = logf(se) + mx - vt;
            const float gs = w * inv_cnt * w_ce / se;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) g[c] = g[c] * gs - ((c == cls) ? w * inv_cnt * w_ce : 0.f);
            // ---- boxes: p = sigmoid(raw) as (cx, cy, w, h)
            const float tg[4] = {tb.x, tb.y, tb.z, tb.w};
            float p[4], dp[4];
            float reg = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                p[k] = 1.0f / (1.0f + expf(-v[NCLS + k]));
                const float df = p[k] - tg[k], ad = fabsf(df);
                const bool quad = ad < beta;
                reg += quad ? 0.5f * df * df / beta : ad - 0.5f * beta;
                dp[k] = (quad ? df / beta : (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f))) * (w_reg * 0.25f);
            }
            // IoU of axis-aligned boxes
            const float ax1 = p[0] - 0.5f * p[2], ax2 = p[0] + 0.5f * p[2];
            const float ay1 = p[1] - 0.5f * p[3], ay2 = p[1] + 0.5f * p[3];
            const float bx1 = tg[0] - 0.5f * tg[2], bx2 = tg[0] + 0.5f * tg[2];
            const float by1 = tg[1] - 0.5f * tg[3], by2 = tg[1] + 0.5f * tg[3];
            const float iw_raw = fminf(ax2, bx2) - fmaxf(ax1, bx1);
            const float ih_raw = fminf(ay2, by2) - fmaxf(ay1, by1);
            const float iw = fmaxf(iw_raw, 0.f), ih = fmaxf(ih_raw, 0.f);
            const float inter = iw * ih;
            const float uni = p[2] * p[3] + tg[2] * tg[3] - inter;
            const float den = uni + iou_eps;
            const float iou = inter / den;
            // d iou / d inter (union depends on inter) and d iou / d area_pred
            const float di_dinter = (den + inter) / (den * den);
            const float di_darea = -inter / (den * den);
            // subgradients of min/max/clamp follow torch: clamp passes at >= 0, min/max pick the selected arm
            const float x2s = (ax2 < bx2) ? 1.f : (ax2 == bx2 ? 0.5f : 0.f);
            const float x1s = (ax1 > bx1) ? 1.f : (ax1 == bx1 ? 0.5f : 0.f);
            const float y2s = (ay2 < by2) ? 1.f : (ay2 == by2 ? 0.5f : 0.f);
            const float y1s = (ay1 > by1) ? 1.f : (ay1 == by1 ? 0.5f : 0.f);
            const float diw = (iw_raw >= 0.f) ? ih * di_dinter : 0.f;   // d iou / d iw
            const float dih = (ih_raw >= 0.f) ? iw * di_dinter : 0.f;
            float di[4];
            di[0] = diw * (x2s - x1s);
            di[1] = dih * (y2s - y1s);
            di[2] = diw * 0.5f * (x2s + x1s) + di_darea * p[3];
            di[3] = dih * 0.5f * (y2s + y1s) + di_darea * p[2];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                g[NCLS + k] = (dp[k] - w_iou * di[k]) * p[k] * (1.f - p[k]) * w * inv_cnt;
#pragma unroll
            for (int c = 0; c < LOSS_F4; ++c)
                st4(row + 4 * c, make_float4(g[4 * c], g[4 * c + 1], g[4 * c + 2], g[4 * c + 3]));
            s_reg += w * reg * 0.25f;
            s_iou += w * (1.f - iou);
            s_ce += w * ce;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LOSS_NV; ++i) {
            const int f = tid + i * LOSS_BLOCK, tok = f / LOSS_F4, c = f - tok * LOSS_F4;
            if (m0 + tok < M) st4(dout + (m0 + tok) * ld + 4 * c, ld4(rows + tok * LOSS_ROWF + 4 * c));
        }
        __syncthreads();                          // the rows are free for the next pass
    }
    s_reg = block_sum(s_reg, red);
    s_iou = block_sum(s_iou, red);
    s_ce = block_sum(s_ce, red);
    if (tid == 0) {
        float* part = scratch + 4 + 4 * blockIdx.x;
        part[0] = s_reg * inv_cnt; part[1] = s_iou * inv_cnt; part[2] = s_ce * inv_cnt; part[3] = 0.f;
        // publish the partial (agent scope), take a ticket; the block that draws the last one sees every partial
        __threadfence();
        int* ticket = reinterpret_cast<int*>(scratch) + 1;
        const int drawn = atomicAdd(ticket, 1);
        last_sh = (drawn == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (last_sh && tid < 64) {
        __threadfence();
        float a = 0.f, b = 0.f, c = 0.f;
        for (int i = tid; i < (int)gridDim.x; i += 64) {           // fixed order: lane i sums blocks i, i + 64, ... ; then the shuffle tree
            const float4 q = ld4(scratch + 4 + 4 * i);
            a += q.x; b += q.y; c += q.z;
        }
        a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
        if (tid == 0) {
            loss_out[0] = w_reg * a + w_iou * b + w_ce * c;
            loss_out[1] = a; loss_out[2] = b; loss_out[3] = c;
            *(reinterpret_cast<int*>(scratch) + 1) = 0;              // ready for the next launch
        }
    }
}

extern "C" int vlg_layout_loss_scratch(void) { return LOSS_SCRATCH; }

extern "C" int vlg_layout_loss(const float* out, int ld, const int64_t* tgt_class, const float* tgt_box,
                               const float* valid, float* dout, float* loss_out, float* scratch, int B, int T,
                               int N, int n_classes, float beta, float iou_eps, float w_reg, float w_iou,
                               float w_ce, void* stream) {
    if (n_classes != NCLS || B < 1 || T < 1 || N < 1 || ld < NOUT || (ld & 3) || !(beta > 0.f)) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(out) || !vlg_aligned16(dout) || !vlg_aligned16(tgt_box) || !vlg_aligned16(valid)) return VLG_ERR_ALIGN;
    const int64_t M = (int64_t)B * T * N;
    if (!loss_out || !scratch || !vlg_aligned16(scratch)) return VLG_ERR_ALIGN;
    int64_t blocks = (M + LOSS_TOK - 1) / LOSS_TOK;
    if (blocks > LOSS_MAX_BLOCKS) blocks = LOSS_MAX_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(layout_loss_kernel, dim3((unsigned)blocks), dim3(LOSS_BLOCK), 0, s, out, ld, tgt_class,
                       tgt_box, valid, dout, scratch, loss_out, B, T, N, beta, iou_eps, w_reg, w_iou, w_ce);
    return vlg_last_error();
}
