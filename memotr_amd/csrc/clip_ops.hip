// clip_ops.hip -- the small-tensor chains of the clip train step as single gfx950 kernels (C ABI: include/clip_ops_hip.h).
//
// Everything here works on a few thousand elements: the cost is the launch, not the arithmetic.  One thread per
// output element, straight-line fp32 code following the reference formulas (operation order kept where it is
// cheap to do so; `#pragma clang fp contract(off)` so that a product rounds before it is added, as in torch's
// separate kernels), fixed-order reductions where a sum is needed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/clip_ops_hip.h"
#include "assign_core.h"

namespace {

thread_local char g_err[256] = {0};

int fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    g_err[0] = 0;
    return 0;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct Box {
    float x1, y1, x2, y2;
};

// cxcywh -> xyxy (reference utils/box_ops.py:16-20)
__device__ __forceinline__ Box to_xyxy(float cx, float cy, float w, float h) {
#pragma clang fp contract(off)
    Box b;
    b.x1 = cx - 0.5f * w;
    b.y1 = cy - 0.5f * h;
    b.x2 = cx + 0.5f * w;
    b.y2 = cy + 0.5f * h;
    return b;
}

// generalised IoU of two xyxy boxes (reference utils/box_ops.py:49-70, one pair)
__device__ __forceinline__ float giou_pair(const Box a, const Box b) {
#pragma clang fp contract(off)
    const float iw = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
    const float ih = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
    const float inter = iw * ih;
    const float area_a = (a.x2 - a.x1) * (a.y2 - a.y1);
    const float area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
    const float uni = area_a + area_b - inter;
    const float iou = inter / uni;
    const float hw = fmaxf(fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1), 0.f);
    const float hh = fmaxf(fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1), 0.f);
    const float hull = hw * hh;
    return iou - (hull - uni) / hull;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float powg(float x, float gamma) { return gamma == 2.f ? x * x : powf(x, gamma); }

// ----------------------------------------------------------------------------------------
// matching cost
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void match_cost_kernel(const float *__restrict__ logits, long lsl, long lsq,
                                                        const float *__restrict__ boxes, long bsl, long bsq,
                                                        const int64_t *__restrict__ gt_labels,
                                                        const float *__restrict__ gt_boxes, int n_layers, int Q,
                                                        int K, int T, float w_class, float w_bbox, float w_giou,
                                                        float *__restrict__ cost) {
#pragma clang fp contract(off)
    const long total = (long)n_layers * Q * T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const long lq = i / T;
        const int q = (int)(lq % Q), l = (int)(lq / Q);
        long lab = gt_labels[t];
        lab = lab < 0 ? 0 : (lab >= K ? K - 1 : lab);
        const float prob = sigmoidf(logits[l * lsl + q * lsq + lab]);
        const float alpha = 0.25f, gamma = 2.f;
        const float neg = ((1.f - alpha) * powg(prob, gamma)) * (-logf(1.f - prob + 1e-8f));
        const float pos = (alpha * powg(1.f - prob, gamma)) * (-logf(prob + 1e-8f));
        const float c_class = pos - neg;
        const float *pb = boxes + l * bsl + q * bsq;
        const float *tb = gt_boxes + (long)t * 4;
        const float p0 = pb[0], p1 = pb[1], p2 = pb[2], p3 = pb[3];
        const float t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
        const float c_bbox = ((fabsf(p0 - t0) + fabsf(p1 - t1)) + fabsf(p2 - t2)) + fabsf(p3 - t3);
        const float giou = giou_pair(to_xyxy(p0, p1, p2, p3), to_xyxy(t0, t1, t2, t3));
        cost[i] = (w_bbox * c_bbox + w_class * c_class) + w_giou * (-giou);
    }
}

// ----------------------------------------------------------------------------------------
// paired box losses
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_box_loss_fwd_kernel(const float *__restrict__ boxes,
                                                               const int64_t *__restrict__ lay,
                                                               const int64_t *__restrict__ qidx, long row_mul,
                                                               long row_add, const float *__restrict__ tgt,
                                                               const int64_t *__restrict__ gidx,
                                                               const float *__restrict__ weight, int n,
                                                               float *__restrict__ l1, float *__restrict__ gl) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *pb = boxes + (lay[i] * row_mul + row_add + qidx[i]) * 4;
    const float *tb = tgt + (gidx ? gidx[i] : (long)i) * 4;
    const float p0 = pb[0], p1 = pb[1], p2 = pb[2], p3 = pb[3];
    const float t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
    float a = ((fabsf(p0 - t0) + fabsf(p1 - t1)) + fabsf(p2 - t2)) + fabsf(p3 - t3);
    float g = 1.f - giou_pair(to_xyxy(p0, p1, p2, p3), to_xyxy(t0, t1, t2, t3));
    if (weight) {
        a *= weight[i];
        g *= weight[i];
    }
    l1[i] = a;
    gl[i] = g;
}

__global__ __launch_bounds__(256) void pair_iou_kernel(const float *__restrict__ boxes, const float *__restrict__ tgt,
                                                      const int64_t *__restrict__ gidx, int n,
                                                      float *__restrict__ iou) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *pb = boxes + (long)i * 4;
    const float *tb = tgt + (gidx ? gidx[i] : (long)i) * 4;
    const Box a = to_xyxy(pb[0], pb[1], pb[2], pb[3]), b = to_xyxy(tb[0], tb[1], tb[2], tb[3]);
    const float iw = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
    const float ih = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
    const float inter = iw * ih;
    const float uni = (a.x2 - a.x1) * (a.y2 - a.y1) + (b.x2 - b.x1) * (b.y2 - b.y1) - inter;
    iou[i] = inter / uni;
}

// d max(a,b)/da and d min(a,b)/da with torch's tie rule (half each)
__device__ __forceinline__ float dmax_a(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float dmin_a(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void pair_box_loss_bwd_kernel(const float *__restrict__ boxes,
                                                               const int64_t *__restrict__ lay,
                                                               const int64_t *__restrict__ qidx, long row_mul,
                                                               long row_add, const float *__restrict__ tgt,
                                                               const int64_t *__restrict__ gidx,
                                                               const float *__restrict__ weight, int n,
                                                               const float *__restrict__ g_l1,
                                                               const float *__restrict__ g_gl,
                                                               float *__restrict__ grad_boxes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long row = lay[i] * row_mul + row_add + qidx[i];
    const float *pb = boxes + row * 4;
    const float *tb = tgt + (gidx ? gidx[i] : (long)i) * 4;
    const float p0 = pb[0], p1 = pb[1], p2 = pb[2], p3 = pb[3];
    const float t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
    const float w = weight ? weight[i] : 1.f;
    const float ga = g_l1[i] * w;         // d/d(l1 sum)
    const float gg = -g_gl[i] * w;        // d/d(giou): the loss is 1 - giou
    const Box a = to_xyxy(p0, p1, p2, p3), b = to_xyxy(t0, t1, t2, t3);
    // forward values
    const float ltx = fmaxf(a.x1, b.x1), lty = fmaxf(a.y1, b.y1), rbx = fminf(a.x2, b.x2), rby = fminf(a.y2, b.y2);
    const float dw = rbx - ltx, dh = rby - lty;
    const float iw = fmaxf(dw, 0.f), ih = fmaxf(dh, 0.f);
    const float inter = iw * ih;
    const float aw = a.x2 - a.x1, ah = a.y2 - a.y1;
    const float area_a = aw * ah, area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
    const float uni = area_a + area_b - inter;
    const float hx1 = fminf(a.x1, b.x1), hy1 = fminf(a.y1, b.y1), hx2 = fmaxf(a.x2, b.x2), hy2 = fmaxf(a.y2, b.y2);
    const float ew = hx2 - hx1, eh = hy2 - hy1;
    const float hw = fmaxf(ew, 0.f), hh = fmaxf(eh, 0.f);
    const float hull = hw * hh;
    // giou = inter/uni - (hull - uni)/hull
    const float g_inter0 = gg / uni;                                          // via iou
    const float g_uni = gg * (-inter / (uni * uni)) + gg / hull;              // via iou and via -(hull-uni)/hull
    const float g_hull = gg * (-(uni / (hull * hull)));                       // d[-(hull-uni)/hull]/d hull = -uni/hull^2
    const float g_inter = g_inter0 - g_uni;                                   // uni = area_a + area_b - inter
    const float g_area_a = g_uni;
    // inter = clamp(dw) * clamp(dh)   (clamp(min=0) passes the gradient where x >= 0)
    const float g_dw = g_inter * ih * (dw >= 0.f ? 1.f : 0.f), g_dh = g_inter * iw * (dh >= 0.f ? 1.f : 0.f);
    const float g_ew = g_hull * hh * (ew >= 0.f ? 1.f : 0.f), g_eh = g_hull * hw * (eh >= 0.f ? 1.f : 0.f);
    // xyxy gradients of box a
    float gx1 = -g_dw * dmax_a(a.x1, b.x1) - g_ew * dmin_a(a.x1, b.x1) - g_area_a * ah;
    float gy1 = -g_dh * dmax_a(a.y1, b.y1) - g_eh * dmin_a(a.y1, b.y1) - g_area_a * aw;
    float gx2 = g_dw * dmin_a(a.x2, b.x2) + g_ew * dmax_a(a.x2, b.x2) + g_area_a * ah;
    float gy2 = g_dh * dmin_a(a.y2, b.y2) + g_eh * dmax_a(a.y2, b.y2) + g_area_a * aw;
    // cxcywh: x1 = cx - w/2, x2 = cx + w/2
    float *go = grad_boxes + row * 4;
    go[0] = (gx1 + gx2) + ga * sgn(p0 - t0);
    go[1] = (gy1 + gy2) + ga * sgn(p1 - t1);
    go[2] = 0.5f * (gx2 - gx1) + ga * sgn(p2 - t2);
    go[3] = 0.5f * (gy2 - gy1) + ga * sgn(p3 - t3);
}

// ----------------------------------------------------------------------------------------
// focal loss of stacked layers
// ----------------------------------------------------------------------------------------
struct Focal {
    float loss, dloss;
};

// one element: logit x, binary target (t == 1 when is_pos)
__device__ __forceinline__ Focal focal_elem(float x, bool is_pos, float alpha, float gamma) {
    const float p = sigmoidf(x);
    // binary_cross_entropy_with_logits: max(x,0) - x*t + log(1 + exp(-|x|))
    const float ce = fmaxf(x, 0.f) - (is_pos ? x : 0.f) + log1pf(expf(-fabsf(x)));
    const float p_t = is_pos ? p : 1.f - p;
    const float q = 1.f - p_t;
    const float a_t = alpha >= 0.f ? (is_pos ? alpha : 1.f - alpha) : 1.f;
    const float mod = powg(q, gamma);
    Focal f;
    f.loss = a_t * (ce * mod);
    // d/dx: -s a_t [gamma q^(gamma-1) (p_t q) ce + q^gamma q],  s = +1 (positive) / -1
    const float qg1 = gamma == 2.f ? q : powf(q, gamma - 1.f);
    const float d = a_t * (gamma * qg1 * (p_t * q) * ce + mod * q);
    f.dloss = is_pos ? -d : d;
    return f;
}

__global__ __launch_bounds__(256) void focal_fwd_kernel(const float *__restrict__ logits, long sl, long sq,
                                                       const int64_t *__restrict__ labels, int Nq, int K, float alpha,
                                                       float gamma, float *__restrict__ loss) {
    __shared__ float s_part[256];
    const int l = blockIdx.x;
    const int total = Nq * K;
    float acc = 0.f;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int q = i / K, k = i - q * K;
        const float x = logits[l * sl + q * sq + k];
        acc += focal_elem(x, labels[(long)l * Nq + q] == k, alpha, gamma).loss;
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_part[threadIdx.x] += s_part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[l] = s_part[0] / (float)K;      // mean over the classes, sum over the queries
}

__global__ __launch_bounds__(256) void focal_bwd_kernel(const float *__restrict__ logits, long sl, long sq,
                                                       const int64_t *__restrict__ labels, int n_layers, int Nq,
                                                       int K, float alpha, float gamma,
                                                       const float *__restrict__ g_loss,
                                                       float *__restrict__ grad_logits) {
    const long total = (long)n_layers * Nq * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long lq = i / K;
        const int q = (int)(lq % Nq), l = (int)(lq / Nq);
        const float x = logits[l * sl + q * sq + k];
        const Focal f = focal_elem(x, labels[lq] == k, alpha, gamma);
        grad_logits[i] = (g_loss[l] / (float)K) * f.dloss;
    }
}

// ----------------------------------------------------------------------------------------
// decoder glue: sine embedding of the anchors, clamped logit, box refinement
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sine_embed_fwd_kernel(const float *__restrict__ pos,
                                                            const float *__restrict__ dim_t, long n, int K, int F,
                                                            float scale, float *__restrict__ out) {
#pragma clang fp contract(off)
    const long total = n * K * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % F);
        const long ik = i / F;                      // = row * K + k
        const float e = (pos[ik] * scale) / dim_t[j];
        out[i] = (j & 1) ? cosf(e) : sinf(e);
    }
}

// one wavefront per (row, coordinate): the F terms of its gradient, summed in a fixed order
__global__ __launch_bounds__(256) void sine_embed_bwd_kernel(const float *__restrict__ pos,
                                                            const float *__restrict__ dim_t, long n, int K, int F,
                                                            float scale, const float *__restrict__ g,
                                                            float *__restrict__ gpos) {
    const int lane = threadIdx.x & 63;
    const long waves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long ik = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; ik < n * K; ik += waves) {
        const float ps = pos[ik] * scale;
        float acc = 0.f;
        for (int j = lane; j < F; j += 64) {
            const float e = ps / dim_t[j];
            const float d = (j & 1) ? -sinf(e) : cosf(e);
            acc += (g[ik * F + j] * d) / dim_t[j];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) gpos[ik] = acc * scale;
    }
}

__device__ __forceinline__ float clamp_keep_nan(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

__device__ __forceinline__ float inv_sigmoid(float x, float eps) {
    return logf(clamp_keep_nan(x, eps, 1.f) / clamp_keep_nan(1.f - x, eps, 1.f));
}

// d inverse_sigmoid / dx with torch's clamp(min, max) masks (the gradient passes where min <= v <= max)
__device__ __forceinline__ float inv_sigmoid_grad(float x, float eps) {
    const float u = 1.f - x;
    const float x1 = clamp_keep_nan(x, eps, 1.f), x2 = clamp_keep_nan(u, eps, 1.f);
    const float a = (x >= eps && x <= 1.f) ? 1.f / x1 : 0.f;
    const float b = (u >= eps && u <= 1.f) ? 1.f / x2 : 0.f;
    return a + b;
}

__global__ __launch_bounds__(256) void inverse_sigmoid_fwd_kernel(const float *__restrict__ x, long n, float eps,
                                                                 float *__restrict__ y) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = inv_sigmoid(x[i], eps);
}

__global__ __launch_bounds__(256) void inverse_sigmoid_bwd_kernel(const float *__restrict__ x,
                                                                 const float *__restrict__ g, long n, float eps,
                                                                 float *__restrict__ gx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        gx[i] = g[i] * inv_sigmoid_grad(x[i], eps);
}

__global__ __launch_bounds__(256) void refine_boxes_fwd_kernel(const float *__restrict__ delta,
                                                              const float *__restrict__ ref, long n, float eps,
                                                              float *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = sigmoidf(delta[i] + inv_sigmoid(ref[i], eps));
}

__global__ __launch_bounds__(256) void refine_boxes_bwd_kernel(const float *__restrict__ out,
                                                              const float *__restrict__ ref,
                                                              const float *__restrict__ g, long n, float eps,
                                                              float *__restrict__ gdelta, float *__restrict__ gref) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float y = out[i];
        const float gs = (g[i] * (1.f - y)) * y;          // sigmoid backward
        gdelta[i] = gs;
        if (gref) gref[i] = gs * inv_sigmoid_grad(ref[i], eps);
    }
}

// 32 columns x 8 row lanes per workgroup; lane (ty, tx) sums rows ty, ty + 8, ... of column tile tx
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, long rows, int cols,
                                                    float *__restrict__ out) {
    __shared__ float s_part[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        long r = ty;
        for (; r + 24 < rows; r += 32) {          // four independent loads in flight
            a0 += x[r * cols + c];
            a1 += x[(r + 8) * cols + c];
            a2 += x[(r + 16) * cols + c];
            a3 += x[(r + 24) * cols + c];
        }
        for (; r < rows; r += 8) a0 += x[r * cols + c];
    }
    s_part[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && c < cols) {
        float t = s_part[0][tx];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += s_part[i][tx];
        out[c] = t;
    }
}

// Short matrices (a few hundred rows) are latency-bound, not bandwidth-bound: one workgroup per FOUR columns, thread t
// takes rows t, t + 256, ... as 16-byte loads (one or two per thread at 310 rows), then a wavefront shuffle reduction
// and one LDS step over the four wavefronts.  6.0 -> ~3 us at 310 x 256 against the 32 x 8 tile above.
__global__ __launch_bounds__(256) void colsum_quad_kernel(const float *__restrict__ x, long rows, int cols,
                                                         float *__restrict__ out) {
    __shared__ f32x4_t s_w[4];
    const int c = blockIdx.x * 4;
    f32x4_t a = {0.f, 0.f, 0.f, 0.f};
    for (long r = threadIdx.x; r < rows; r += 256) a += *reinterpret_cast<const f32x4_t *>(x + r * cols + c);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a.x += __shfl_xor(a.x, o, 64);
        a.y += __shfl_xor(a.y, o, 64);
        a.z += __shfl_xor(a.z, o, 64);
        a.w += __shfl_xor(a.w, o, 64);
    }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<f32x4_t *>(out + c) = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// the same tile over one chunk of rows per workgroup row (blockIdx.y): partial sums for tall matrices
__device__ __forceinline__ float colsum_ld(const float *x, long i) { return x[i]; }
__device__ __forceinline__ float colsum_ld(const uint16_t *x, long i) {      // bf16 storage, fp32 sums
    return __uint_as_float(((unsigned)x[i]) << 16);
}

template <typename TX>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const TX *__restrict__ x, long rows, int cols,
                                                            int chunk_rows, float *__restrict__ partial) {
    __shared__ float s_part[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const long r0 = (long)blockIdx.y * chunk_rows;
    const long r1 = r0 + chunk_rows < rows ? r0 + chunk_rows : rows;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        long r = r0 + ty;
        for (; r + 24 < r1; r += 32) {
            a0 += colsum_ld(x, r * cols + c);
            a1 += colsum_ld(x, (r + 8) * cols + c);
            a2 += colsum_ld(x, (r + 16) * cols + c);
            a3 += colsum_ld(x, (r + 24) * cols + c);
        }
        for (; r < r1; r += 8) a0 += colsum_ld(x, r * cols + c);
    }
    s_part[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && c < cols) {
        float t = s_part[0][tx];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += s_part[i][tx];
        partial[(long)blockIdx.y * cols + c] = t;
    }
}

// The ReLU mask of a fused-ReLU Linear's backward and the first pass of its bias gradient in ONE pass over the
// gradient: g2 = y > 0 ? g : 0 is written and summed per (32-column tile, row chunk) on the way.  For the encoder
// FFN's first linear (111,615 x 2048 floats per layer of a 5-frame clip) the separate column sum re-read 914 MB.
__device__ __forceinline__ void colsum_st(float *x, long i, float v) { x[i] = v; }

template <typename TX>
__global__ __launch_bounds__(256) void relu_bwd_colsum_partial_kernel(const TX *__restrict__ g, const TX *__restrict__ y,
                                                                     long rows, int cols, int chunk_rows,
                                                                     TX *__restrict__ g2, float *__restrict__ partial) {
    __shared__ float s_part[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const long r0 = (long)blockIdx.y * chunk_rows;
    const long r1 = r0 + chunk_rows < rows ? r0 + chunk_rows : rows;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        long r = r0 + ty;
        for (; r + 24 < r1; r += 32) {
            const long i0 = r * cols + c, i1 = (r + 8) * cols + c, i2 = (r + 16) * cols + c, i3 = (r + 24) * cols + c;
            const float v0 = colsum_ld(y, i0) <= 0.f ? 0.f : colsum_ld(g, i0);
            const float v1 = colsum_ld(y, i1) <= 0.f ? 0.f : colsum_ld(g, i1);
            const float v2 = colsum_ld(y, i2) <= 0.f ? 0.f : colsum_ld(g, i2);
            const float v3 = colsum_ld(y, i3) <= 0.f ? 0.f : colsum_ld(g, i3);
            colsum_st(g2, i0, v0); colsum_st(g2, i1, v1); colsum_st(g2, i2, v2); colsum_st(g2, i3, v3);
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; r < r1; r += 8) {
            const long i0 = r * cols + c;
            const float v0 = colsum_ld(y, i0) <= 0.f ? 0.f : colsum_ld(g, i0);
            colsum_st(g2, i0, v0);
            a0 += v0;
        }
    }
    s_part[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && c < cols) {
        float t = s_part[0][tx];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += s_part[i][tx];
        partial[(long)blockIdx.y * cols + c] = t;
    }
}

// ----------------------------------------------------------------------------------------
// multi-head self-attention over the decoder queries (head_dim 32, L <= 512): forward and backward
// ----------------------------------------------------------------------------------------
// One workgroup = kRows rows x kLanes lanes (256 threads) of one (batch, head).  A lane of a row visits the rows of
// the OTHER axis j = lane, lane + kLanes, ... : 32-float dot products against rows staged in LDS (pitch 36 floats: the
// sixteen lanes of a row cover all 64 banks exactly once per ds_read_b128, the four rows of a wavefront read the same
// addresses = broadcast).  16 lanes per row: a decoder call is only ~2,500 rows x 320 keys, so the per-lane chain
// (20 keys) sets the kernel's latency, not throughput.
constexpr int kLanes = 16;         // lanes per row
constexpr int kRows = 256 / kLanes;  // rows per workgroup
constexpr int kHd = 32;            // head dimension
constexpr int kPitch = 36;         // LDS row pitch in floats

__device__ __forceinline__ void load_row32(const float *p, float *r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t *>(p + 4 * i);
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
}

__device__ __forceinline__ float dot32(const float *a, const float *lds_row) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t *>(lds_row + 4 * i);
        s0 = fmaf(a[4 * i], v.x, s0);
        s1 = fmaf(a[4 * i + 1], v.y, s1);
        s0 = fmaf(a[4 * i + 2], v.z, s0);
        s1 = fmaf(a[4 * i + 3], v.w, s1);
    }
    return s0 + s1;
}

__device__ __forceinline__ void axpy32(float *acc, float w, const float *lds_row) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t *>(lds_row + 4 * i);
        acc[4 * i] = fmaf(w, v.x, acc[4 * i]);
        acc[4 * i + 1] = fmaf(w, v.y, acc[4 * i + 1]);
        acc[4 * i + 2] = fmaf(w, v.z, acc[4 * i + 2]);
        acc[4 * i + 3] = fmaf(w, v.w, acc[4 * i + 3]);
    }
}

// sum / max over the kLanes lanes of a row (they differ in the low bits of the lane id)
__device__ __forceinline__ float sum8l(float x) {
#pragma unroll
    for (int o = 1; o < kLanes; o <<= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float max8l(float x) {
#pragma unroll
    for (int o = 1; o < kLanes; o <<= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

// cooperative copy of L rows (32 floats each, `rs` apart) into LDS at pitch kPitch
__device__ __forceinline__ void stage_rows(float *dst, const float *src, long rs, int L) {
    for (int i = threadIdx.x; i < L * 8; i += blockDim.x) {
        const int r = i >> 3, c = (i & 7) * 4;
        *reinterpret_cast<f32x4_t *>(dst + r * kPitch + c) = *reinterpret_cast<const f32x4_t *>(src + r * rs + c);
    }
}

__global__ __launch_bounds__(256) void mha_fwd_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                     const float *__restrict__ v, long q_bs, long q_rs, long k_bs,
                                                     long k_rs, long v_bs, long v_rs,
                                                     const uint8_t *__restrict__ key_mask, int H, int L, float scale,
                                                     float *__restrict__ out, float *__restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float *Ks = s_mem, *Vs = s_mem + (size_t)L * kPitch;
    const int h = blockIdx.y, b = blockIdx.z;
    stage_rows(Ks, k + b * k_bs + h * kHd, k_rs, L);
    stage_rows(Vs, v + b * v_bs + h * kHd, v_rs, L);
    __syncthreads();
    const int row = blockIdx.x * kRows + threadIdx.x / kLanes, sub = threadIdx.x % kLanes;
    const bool row_ok = row < L;
    const int rowc = row_ok ? row : L - 1;
    float qr[kHd];
    load_row32(q + b * q_bs + rowc * q_rs + h * kHd, qr);
#pragma unroll
    for (int i = 0; i < kHd; ++i) qr[i] *= scale;
    const uint8_t *mk = key_mask ? key_mask + (long)b * L : nullptr;
    float m = -INFINITY, l = 0.f, acc[kHd];
#pragma unroll
    for (int i = 0; i < kHd; ++i) acc[i] = 0.f;
    for (int j = sub; j < L; j += kLanes) {
        if (mk && mk[j]) continue;
        const float s = dot32(qr, Ks + j * kPitch);
        const float m_new = fmaxf(m, s);
        const float corr = expf(m - m_new), p = expf(s - m_new);      // exp(-inf) = 0 on the first key
        l = l * corr + p;
#pragma unroll
        for (int i = 0; i < kHd; ++i) acc[i] *= corr;
        axpy32(acc, p, Vs + j * kPitch);
        m = m_new;
    }
    // merge the lanes of the row
    const float M = max8l(m);
    const float w = (m == -INFINITY) ? 0.f : expf(m - M);          // a lane may have seen no key at all
    const float Lsum = sum8l(l * w);
    const float inv = 1.f / Lsum;
    float *o = out + ((long)b * L + rowc) * (H * kHd) + h * kHd;
#pragma unroll
    for (int i = 0; i < kHd; ++i) {
        const float t = sum8l(acc[i] * w) * inv;
        if (row_ok && (i >> 2) == sub) o[i] = t;                   // lane `sub` stores floats 4*sub .. 4*sub+3
    }
    if (row_ok && sub == 0) lse[((long)b * H + h) * L + row] = M + logf(Lsum);
}

// grad_q: same decomposition as the forward (a workgroup = kRows query rows; K and V of the head in LDS)
__global__ __launch_bounds__(256) void mha_bwd_q_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                       const float *__restrict__ v, long q_bs, long q_rs, long k_bs,
                                                       long k_rs, long v_bs, long v_rs,
                                                       const uint8_t *__restrict__ key_mask,
                                                       const float *__restrict__ out, const float *__restrict__ lse,
                                                       const float *__restrict__ go, int H, int L, float scale,
                                                       float *__restrict__ gq, long gq_bs, long gq_rs) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float *Ks = s_mem, *Vs = s_mem + (size_t)L * kPitch;
    const int h = blockIdx.y, b = blockIdx.z;
    stage_rows(Ks, k + b * k_bs + h * kHd, k_rs, L);
    stage_rows(Vs, v + b * v_bs + h * kHd, v_rs, L);
    __syncthreads();
    const int row = blockIdx.x * kRows + threadIdx.x / kLanes, sub = threadIdx.x % kLanes;
    const bool row_ok = row < L;
    const int rowc = row_ok ? row : L - 1;
    float qr[kHd], gr[kHd], acc[kHd];
    load_row32(q + b * q_bs + rowc * q_rs + h * kHd, qr);
    const long orow = ((long)b * L + rowc) * (H * kHd) + h * kHd;
    load_row32(go + orow, gr);
    float D = 0.f;                                                  // rowsum(dP o P) = dO . O
    {
        float orr[kHd];
        load_row32(out + orow, orr);
#pragma unroll
        for (int i = 0; i < kHd; ++i) D = fmaf(gr[i], orr[i], D);
    }
#pragma unroll
    for (int i = 0; i < kHd; ++i) { qr[i] *= scale; acc[i] = 0.f; }
    const float ls = lse[((long)b * H + h) * L + rowc];
    const uint8_t *mk = key_mask ? key_mask + (long)b * L : nullptr;
    for (int j = sub; j < L; j += kLanes) {
        if (mk && mk[j]) continue;
        const float p = expf(dot32(qr, Ks + j * kPitch) - ls);
        const float ds = p * (dot32(gr, Vs + j * kPitch) - D);
        axpy32(acc, ds, Ks + j * kPitch);
    }
    float *g = gq + b * gq_bs + rowc * gq_rs + h * kHd;
#pragma unroll
    for (int i = 0; i < kHd; ++i) {
        const float t = sum8l(acc[i]) * scale;
        if (row_ok && (i >> 2) == sub) g[i] = t;
    }
}

// grad_k, grad_v: a workgroup = kRows KEY rows; Q and dO of the head in LDS, lse and D = dO . O per query too
__global__ __launch_bounds__(256) void mha_bwd_kv_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                        const float *__restrict__ v, long q_bs, long q_rs, long k_bs,
                                                        long k_rs, long v_bs, long v_rs,
                                                        const uint8_t *__restrict__ key_mask,
                                                        const float *__restrict__ out, const float *__restrict__ lse,
                                                        const float *__restrict__ go, int H, int L, float scale,
                                                        float *__restrict__ gk, long gk_bs, long gk_rs,
                                                        float *__restrict__ gv, long gv_bs, long gv_rs) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float *Qs = s_mem, *Gs = s_mem + (size_t)L * kPitch;
    float *s_lse = Gs + (size_t)L * kPitch, *s_D = s_lse + L;
    const int h = blockIdx.y, b = blockIdx.z;
    stage_rows(Qs, q + b * q_bs + h * kHd, q_rs, L);
    stage_rows(Gs, go + (long)b * L * (H * kHd) + h * kHd, (long)H * kHd, L);
    for (int i = threadIdx.x; i < L; i += blockDim.x) s_lse[i] = lse[((long)b * H + h) * L + i];
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += blockDim.x) {             // D_i = dO_i . O_i
        const float *o = out + ((long)b * L + i) * (H * kHd) + h * kHd;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < kHd; ++c) d = fmaf(Gs[i * kPitch + c], o[c], d);
        s_D[i] = d;
    }
    __syncthreads();
    const int row = blockIdx.x * kRows + threadIdx.x / kLanes, sub = threadIdx.x % kLanes;
    const bool row_ok = row < L;
    const int rowc = row_ok ? row : L - 1;
    const bool dead = key_mask && key_mask[(long)b * L + rowc];     // a masked key receives no gradient
    float kr[kHd], vr[kHd], ak[kHd], av[kHd];
    load_row32(k + b * k_bs + rowc * k_rs + h * kHd, kr);
    load_row32(v + b * v_bs + rowc * v_rs + h * kHd, vr);
#pragma unroll
    for (int i = 0; i < kHd; ++i) { kr[i] *= scale; ak[i] = 0.f; av[i] = 0.f; }
    if (!dead) {
        for (int i = sub; i < L; i += kLanes) {
            const float p = expf(dot32(kr, Qs + i * kPitch) - s_lse[i]);
            const float ds = p * (dot32(vr, Gs + i * kPitch) - s_D[i]);
            axpy32(av, p, Gs + i * kPitch);
            axpy32(ak, ds, Qs + i * kPitch);
        }
    }
    float *pk = gk + b * gk_bs + rowc * gk_rs + h * kHd, *pv = gv + b * gv_bs + rowc * gv_rs + h * kHd;
#pragma unroll
    for (int i = 0; i < kHd; ++i) {
        const float tk = sum8l(ak[i]) * scale, tv = sum8l(av[i]);
        if (row_ok && (i >> 2) == sub) {
            pk[i] = tk;
            pv[i] = tv;
        }
    }
}

// dynamic LDS above 64 KiB has to be allowed once per kernel and device (one bit per device ordinal in `done`)
int allow_lds(const void *kernel, size_t lds, std::atomic<unsigned long long> &done) {
    if (lds <= 64 * 1024) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

std::atomic<unsigned long long> g_lds_fwd{0}, g_lds_bq{0}, g_lds_bkv{0};

int mha_check(const void *q, const void *k, const void *v, int B, int H, int L) {
    if (B < 0 || H <= 0 || L < 0) return fail(1, "clipops_mha: bad dimension");
    if (L > CLIPOPS_MHA_MAX_L) return fail(2, "clipops_mha: L exceeds CLIPOPS_MHA_MAX_L");
    if ((long)B * L > 0 && (!q || !k || !v)) return fail(1, "clipops_mha: null pointer");
    return 0;
}

// ----------------------------------------------------------------------------------------
// residual add + LayerNorm over rows of 256 floats: one wavefront per row, 4 floats per lane
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum64(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__global__ __launch_bounds__(256) void add_layer_norm_fwd_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ res,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, long rows, float eps,
                                                                float *__restrict__ sum, float *__restrict__ y,
                                                                float *__restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long off = row * CLIPOPS_LN_COLS + lane * 4;
    const f32x4_t a = *reinterpret_cast<const f32x4_t *>(x + off);
    const f32x4_t b = *reinterpret_cast<const f32x4_t *>(res + off);
    const f32x4_t s = a + b;
    *reinterpret_cast<f32x4_t *>(sum + off) = s;
    const float mean = wave_sum64((s.x + s.y) + (s.z + s.w)) * (1.f / CLIPOPS_LN_COLS);
    const f32x4_t d = s - mean;
    const float var = wave_sum64((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.f / CLIPOPS_LN_COLS);
    const float rstd = rsqrtf(var + eps);
    const f32x4_t g = *reinterpret_cast<const f32x4_t *>(gamma + lane * 4);
    const f32x4_t be = *reinterpret_cast<const f32x4_t *>(beta + lane * 4);
    *reinterpret_cast<f32x4_t *>(y + off) = (d * rstd) * g + be;
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// a workgroup walks chunk_rows rows (4 at a time, one per wavefront) and keeps the column sums of its rows
__global__ __launch_bounds__(256) void add_layer_norm_bwd_kernel(const float *__restrict__ gy,
                                                                const float *__restrict__ sum,
                                                                const float *__restrict__ stats,
                                                                const float *__restrict__ gamma, long rows,
                                                                int chunk_rows, float *__restrict__ gsum,
                                                                float *__restrict__ partial) {
    __shared__ f32x4_t s_acc[4][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4_t gm = *reinterpret_cast<const f32x4_t *>(gamma + lane * 4);
    f32x4_t acc_g = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
    const long r0 = (long)blockIdx.x * chunk_rows;
    const long r1 = r0 + chunk_rows < rows ? r0 + chunk_rows : rows;
    for (long row = r0 + wave; row < r1; row += 4) {
        const long off = row * CLIPOPS_LN_COLS + lane * 4;
        const f32x4_t g = *reinterpret_cast<const f32x4_t *>(gy + off);
        const f32x4_t s = *reinterpret_cast<const f32x4_t *>(sum + off);
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        const f32x4_t xh = (s - mean) * rstd;
        const f32x4_t gh = g * gm;
        const float c1 = wave_sum64((gh.x + gh.y) + (gh.z + gh.w)) * (1.f / CLIPOPS_LN_COLS);
        const float c2 = wave_sum64((gh.x * xh.x + gh.y * xh.y) + (gh.z * xh.z + gh.w * xh.w)) * (1.f / CLIPOPS_LN_COLS);
        *reinterpret_cast<f32x4_t *>(gsum + off) = (gh - c1 - xh * c2) * rstd;
        acc_g += g * xh;
        acc_b += g;
    }
    s_acc[wave][0][lane] = acc_g;
    s_acc[wave][1][lane] = acc_b;
    __syncthreads();
    if (wave < 2) {                       // wave 0: gamma part, wave 1: beta part
        const f32x4_t t = (s_acc[0][wave][lane] + s_acc[1][wave][lane]) + (s_acc[2][wave][lane] + s_acc[3][wave][lane]);
        *reinterpret_cast<f32x4_t *>(partial + (long)blockIdx.x * 2 * CLIPOPS_LN_COLS + wave * CLIPOPS_LN_COLS + lane * 4) = t;
    }
}

// ---- linear sum assignment on the device (assign_core.h; one wavefront per problem) ----------------------------
// The wavefront's side of assign::Lanes: lane l scans positions l, l + 64, ... of the unvisited-column list and the
// 64 partial results meet in an xor butterfly of the (commutative, associative) merge; everything that is not the scan
// is wave-uniform control flow around LDS arrays, with a workgroup barrier (one wavefront: an s_barrier and the LDS
// wait) wherever one lane's LDS write must be seen by the others.
struct WaveLanes {
    int lane;
    template <typename F>
    __device__ __forceinline__ assign::ScanBest scan(int n, F &&body) const {
        assign::ScanBest mine = assign::scan_empty();
        for (int it = lane; it < n; it += 64) body(it, mine);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            assign::ScanBest o;
            o.lowest = __shfl_xor(mine.lowest, off, 64);
            o.last_free = __shfl_xor(mine.last_free, off, 64);
            o.first = __shfl_xor(mine.first, off, 64);
            mine = assign::scan_merge(mine, o);
        }
        return mine;
    }
    template <typename F>
    __device__ __forceinline__ void each(int n, F &&body) const {
        for (int k = lane; k < n; k += 64) body(k);
    }
    __device__ __forceinline__ bool leader() const { return lane == 0; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
};

__global__ __launch_bounds__(64) void assign_kernel(const float *__restrict__ cost, long stride_p, long stride_r,
                                                    long stride_c, int n_rows, int n_cols,
                                                    int32_t *__restrict__ row_ind, int32_t *__restrict__ col_ind,
                                                    int32_t *__restrict__ status) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_assign[];
    const long p = blockIdx.x;
    const int k = n_rows < n_cols ? n_rows : n_cols;
    const WaveLanes lanes{(int)threadIdx.x};
    const int rc = assign::solve_problem(lanes, cost + p * stride_p, stride_r, stride_c, n_rows, n_cols, s_assign,
                                         row_ind + p * k, col_ind + p * k);
    if (threadIdx.x == 0 && status != nullptr) status[p] = rc;
}

int grid_for(long total) {
    long g = (total + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > 4096 ? 4096 : g);
}


// ---------------------------------------------------------------------------------------------------------------
// Frozen-BN shift + (residual) + ReLU of a ResNet bottleneck in ONE pass over the convolution's output, in place.
// The backbone (reference models/backbone.py:70-76: torchvision's resnet50 with FrozenBatchNorm2d) does this as three
// element-wise passes over activations of up to 344 MB each (5 frames x 256 x 200 x 336): conv bias, `out += identity`,
// ReLU.  NCHW: one (image, channel) plane per blockIdx.y, so the shift is a scalar of the block.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sr_bf16(uint16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ uint16_t sr_to_bf16(float f) {          // round to nearest even (NaN stays NaN)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool RES>
__global__ __launch_bounds__(256) void shift_relu_f32_kernel(float *__restrict__ x, const float *__restrict__ shift,
                                                             const float *__restrict__ res, long planes, int C, long HW,
                                                             int vec) {
    for (long plane = blockIdx.y; plane < planes; plane += gridDim.y) {
        const float b = shift[plane % C];
        float *xp = x + plane * HW;
        const float *rp = RES ? res + plane * HW : nullptr;
        if (vec) {
            const long n4 = HW >> 2;
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
                float4 v = reinterpret_cast<float4 *>(xp)[i];
                if (RES) {
                    const float4 r = reinterpret_cast<const float4 *>(rp)[i];
                    v.x = (v.x + b) + r.x; v.y = (v.y + b) + r.y; v.z = (v.z + b) + r.z; v.w = (v.w + b) + r.w;
                } else {
                    v.x += b; v.y += b; v.z += b; v.w += b;
                }
                // (max(v, 0) would turn NaN into 0; torch's relu keeps it: v < 0 ? 0 : v)
                v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y;
                v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
                reinterpret_cast<float4 *>(xp)[i] = v;
            }
        } else {
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
                float v = xp[i] + b;
                if (RES) v += rp[i];
                xp[i] = v < 0.f ? 0.f : v;
            }
        }
    }
}

// bf16 storage (the autocast step): the bias add and the residual add round to bf16 one after the other, like the two
// torch kernels they replace
template <bool RES>
__global__ __launch_bounds__(256) void shift_relu_bf16_kernel(uint16_t *__restrict__ x, const float *__restrict__ shift,
                                                              const uint16_t *__restrict__ res, long planes, int C,
                                                              long HW, int vec) {
    for (long plane = blockIdx.y; plane < planes; plane += gridDim.y) {
        const float b = sr_bf16(sr_to_bf16(shift[plane % C]));          // autocast hands the convolution a bf16 bias
        uint16_t *xp = x + plane * HW;
        const uint16_t *rp = RES ? res + plane * HW : nullptr;
        if (vec) {
            const long n8 = HW >> 3;
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
                uint4 v = reinterpret_cast<uint4 *>(xp)[i];
                uint4 r = {0u, 0u, 0u, 0u};
                if (RES) r = reinterpret_cast<const uint4 *>(rp)[i];
                unsigned *vw = reinterpret_cast<unsigned *>(&v);
                const unsigned *rw = reinterpret_cast<const unsigned *>(&r);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float lo = sr_bf16((uint16_t)(vw[k] & 0xffffu)) + b, hi = sr_bf16((uint16_t)(vw[k] >> 16)) + b;
                    if (RES) {
                        lo = sr_bf16(sr_to_bf16(lo)) + sr_bf16((uint16_t)(rw[k] & 0xffffu));
                        hi = sr_bf16(sr_to_bf16(hi)) + sr_bf16((uint16_t)(rw[k] >> 16));
                    }
                    lo = lo < 0.f ? 0.f : lo;
                    hi = hi < 0.f ? 0.f : hi;
                    vw[k] = (unsigned)sr_to_bf16(lo) | ((unsigned)sr_to_bf16(hi) << 16);
                }
                reinterpret_cast<uint4 *>(xp)[i] = v;
            }
        } else {
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
                float v = sr_bf16(xp[i]) + b;
                if (RES) v = sr_bf16(sr_to_bf16(v)) + sr_bf16(rp[i]);
                xp[i] = sr_to_bf16(v < 0.f ? 0.f : v);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Backward of a query-sized Linear (a few hundred rows) in ONE launch: grad_x = G' W, grad_w = G'^T X, grad_b = colsum G',
// G' = grad_y (masked by y > 0 when the forward fused a ReLU).  torch issues [threshold_backward,] two GEMMs and a
// reduction -- four dependent launches of ~5 us each inside the decoder's hipGraphs for ~0.1 GFLOP; the three results
// depend on the same inputs only, so they are tiles of one grid here.
//   * one 32 x 32 output tile per workgroup, fp32 MFMA 32x32x2 (bit-for-bit a k-ordered fmaf chain), the contraction
//     split over the four wavefronts and added up through LDS;
//   * lane l feeds A[i = l & 31][k] and B[k][j = l & 31] for k = 8 c + 4 (l >> 5) + {0..3}: four MFMAs per 8 k;
//   * operands straight from global memory (the whole problem sits in L2), zero-filled past the edges.
// ---------------------------------------------------------------------------------------------------------------
typedef float lb_f32x16 __attribute__((ext_vector_type(16)));

// One workgroup's tile.  GX: C = G' W (A rows are rows of G: one 16-byte load per lane and chunk when OUT % 4 == 0);
// else C = G'^T X (A columns are rows of G: coalesced dwords).  LB_U chunks of 8 k are requested before the first MFMA
// of the group -- without that every chunk waits a memory round trip and the kernel takes 40 us at K = 1024.
#ifndef MSDA_LB_U
#define MSDA_LB_U 4
#endif
constexpr int LB_U = MSDA_LB_U;

template <bool GX, bool RELU>
__device__ __forceinline__ void linear_bwd_tile(const float *__restrict__ g, const float *__restrict__ y,
                                                const float *__restrict__ Bp, int M, int N, int K, int OUT, int i0, int j0,
                                                float *__restrict__ C, float *__restrict__ gb, float (*s_c)[16][64],
                                                float (*s_b)[32]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = i0 + (lane & 31), j = j0 + (lane & 31), kq = (lane >> 5) * 4;
    const bool iok = i < M, jok = j < N;
    const int chunks = (K + 7) / 8, per = (chunks + 3) / 4;
    const int c_lo = wave * per, c_hi = (c_lo + per) < chunks ? (c_lo + per) : chunks;
    const bool vec = GX && (OUT % 4 == 0);
    lb_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int c = c_lo; c < c_hi; c += LB_U) {
        float a[LB_U][4], b[LB_U][4], m[LB_U][4];
#pragma unroll
        for (int u = 0; u < LB_U; ++u) {
            const int k0 = (c + u) * 8 + kq;
            const bool cok = c + u < c_hi;
            if (vec) {
                float4 av = {0.f, 0.f, 0.f, 0.f}, mv = {1.f, 1.f, 1.f, 1.f};
                if (cok && iok && k0 < K) {          // (K = OUT is a multiple of 4 and k0 one too: all four or none)
                    av = *reinterpret_cast<const float4 *>(g + (long)i * OUT + k0);
                    if (RELU) mv = *reinterpret_cast<const float4 *>(y + (long)i * OUT + k0);
                }
                a[u][0] = av.x; a[u][1] = av.y; a[u][2] = av.z; a[u][3] = av.w;
                m[u][0] = mv.x; m[u][1] = mv.y; m[u][2] = mv.z; m[u][3] = mv.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + q;
                const bool kok = cok && k < K;
                if (!vec) {
                    const long idx = GX ? (long)i * OUT + k : (long)k * OUT + i;
                    a[u][q] = (iok && kok) ? g[idx] : 0.f;
                    m[u][q] = (RELU && iok && kok) ? y[idx] : 1.f;
                }
                b[u][q] = (jok && kok) ? Bp[(long)k * N + j] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < LB_U; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float av = RELU ? (m[u][q] <= 0.f ? 0.f : a[u][q]) : a[u][q];      // (y <= 0 ? 0 : g -- aten's threshold_backward: a NaN activation passes g)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][q], acc, 0, 0, 0);
                bsum += av;
            }
        }
    }
    // add the four partial tiles (and bias sums) up; wavefront v finishes accumulator registers 4v .. 4v + 3
#pragma unroll
    for (int r = 0; r < 16; ++r) s_c[wave][r][lane] = acc[r];
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 32) s_b[wave][lane] = bsum;
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr;
        const float v = (s_c[0][r][lane] + s_c[1][r][lane]) + (s_c[2][r][lane] + s_c[3][r][lane]);
        const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && jok) C[(long)row * N + j] = v;
    }
    if (!GX && gb != nullptr && j0 == 0 && wave == 0 && lane < 32 && i0 + lane < M)
        gb[i0 + lane] = (s_b[0][lane] + s_b[1][lane]) + (s_b[2][lane] + s_b[3][lane]);
}

template <bool RELU>
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                         const float *__restrict__ x, const float *__restrict__ w,
                                                         int R, int IN, int OUT, float *__restrict__ gx,
                                                         float *__restrict__ gw, float *__restrict__ gb, int n_gx,
                                                         int gx_tj, int gw_tj) {
    __shared__ float s_c[4][16][64];
    __shared__ float s_b[4][32];
    const int tile = blockIdx.x;
    // C (M x N) = A (M x K) B (K x N):   gx: A = G' (R x OUT), B = W (OUT x IN)     gw: A = G'^T (OUT x R), B = X (R x IN)
    if (tile < n_gx) {
        linear_bwd_tile<true, RELU>(g, y, w, R, IN, OUT, OUT, (tile / gx_tj) * 32, (tile % gx_tj) * 32, gx, nullptr, s_c, s_b);
    } else {
        const int t = tile - n_gx;
        linear_bwd_tile<false, RELU>(g, y, x, OUT, IN, R, OUT, (t / gw_tj) * 32, (t % gw_tj) * 32, gw, gb, s_c, s_b);
    }
}

// Forward of the same linears: y = [relu](x W^T + b), one 32 x 32 tile per workgroup.  Both operands are contiguous
// along the contraction (x rows, W rows), so a lane's four k of a chunk are ONE 16-byte load each.
template <bool RELU>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, int R, int IN, int OUT,
                                                         float *__restrict__ yout, int tj) {
    __shared__ float s_c[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x / tj) * 32, j0 = (blockIdx.x % tj) * 32;
    const int i = i0 + (lane & 31), j = j0 + (lane & 31), kq = (lane >> 5) * 4;
    const bool iok = i < R, jok = j < OUT;
    const int chunks = (IN + 7) / 8, per = (chunks + 3) / 4;
    const int c_lo = wave * per, c_hi = (c_lo + per) < chunks ? (c_lo + per) : chunks;
    lb_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = c_lo; c < c_hi; c += LB_U) {
        float4 a[LB_U], b[LB_U];
#pragma unroll
        for (int u = 0; u < LB_U; ++u) {
            const int k0 = (c + u) * 8 + kq;
            const bool kok = c + u < c_hi && k0 < IN;          // (IN is a multiple of 4: the launcher checks)
            a[u] = (iok && kok) ? *reinterpret_cast<const float4 *>(x + (long)i * IN + k0) : float4{0.f, 0.f, 0.f, 0.f};
            b[u] = (jok && kok) ? *reinterpret_cast<const float4 *>(w + (long)j * IN + k0) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < LB_U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_c[wave][r][lane] = acc[r];
    __syncthreads();
    const float bj = (bias != nullptr && jok) ? bias[j] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr;
        float v = ((s_c[0][r][lane] + s_c[1][r][lane]) + (s_c[2][r][lane] + s_c[3][r][lane])) + bj;
        if (RELU) v = v < 0.f ? 0.f : v;
        const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < R && jok) yout[(long)row * OUT + j] = v;
    }
}

}  // namespace

// ---- per-frame bookkeeping of the criterion (round 6, ABI 10): index arithmetic on a few hundred elements that took
//      7 and 11 torch launches per frame -- the forward between two decoder graphs is bound by the host's launch rate ----
// one workgroup: thread i < n_tr finds the LAST ground truth carrying track i's id, thread j < n_gt whether any track
// carries ground truth j's id
__global__ __launch_bounds__(256) void track_ownership_kernel(const int64_t *__restrict__ tr_ids, int n_tr,
                                                              const int64_t *__restrict__ gt_ids, int n_gt,
                                                              int64_t *__restrict__ matched, float *__restrict__ free_out) {
    for (int i = threadIdx.x; i < n_tr; i += blockDim.x) {
        const int64_t id = tr_ids[i];
        int64_t m = -1;
        for (int j = 0; j < n_gt; ++j)
            if (gt_ids[j] == id) m = j;
        matched[i] = m;
    }
    for (int j = threadIdx.x; j < n_gt; j += blockDim.x) {
        const int64_t id = gt_ids[j];
        bool owned = false;
        for (int i = 0; i < n_tr; ++i) owned = owned || tr_ids[i] == id;
        free_out[j] = owned ? 0.f : 1.f;
    }
}

// one workgroup: background everywhere, the carried tracks' labels in the late layers, then the matched pairs
__global__ __launch_bounds__(256) void focal_labels_kernel(const int64_t *__restrict__ lay, const int64_t *__restrict__ q,
                                                           const int64_t *__restrict__ g, int n_pairs,
                                                           const int64_t *__restrict__ gt_labels, int n_gt,
                                                           const int64_t *__restrict__ matched, int n_tr,
                                                           const unsigned char *__restrict__ late, int n_layers, int nd,
                                                           int K, int64_t *__restrict__ labels) {
    const int nq = nd + n_tr;
    for (int e = threadIdx.x; e < n_layers * nq; e += blockDim.x) {
        const int l = e / nq, c = e - l * nq;
        int64_t v = K;
        if (c >= nd && late[l]) {
            const int64_t m = matched[c - nd];
            if (m >= 0 && m < n_gt) v = gt_labels[m];
        }
        labels[e] = v;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n_pairs; p += blockDim.x) {
        const int64_t l = lay[p], c = q[p], t = g[p];
        if (l >= 0 && l < n_layers && c >= 0 && c < nq && t >= 0 && t < n_gt) labels[l * nq + c] = gt_labels[t];
    }
}

extern "C" {

int clipops_abi_version(void) { return CLIPOPS_ABI_VERSION; }

const char *clipops_last_error(void) { return g_err; }

int clipops_match_cost_f32(const float *logits, long logit_sl, long logit_sq, const float *boxes, long box_sl,
                           long box_sq, const int64_t *gt_labels, const float *gt_boxes, int n_layers, int Q, int K,
                           int T, float w_class, float w_bbox, float w_giou, float *cost, void *stream) {
    if (n_layers < 0 || Q < 0 || T < 0 || K <= 0) return fail(1, "clipops_match_cost_f32: bad dimension");
    if ((long)n_layers * Q * T == 0) { g_err[0] = 0; return 0; }
    if (!logits || !boxes || !gt_labels || !gt_boxes || !cost) return fail(1, "clipops_match_cost_f32: null pointer");
    hipLaunchKernelGGL(match_cost_kernel, dim3(grid_for((long)n_layers * Q * T)), dim3(256), 0, (hipStream_t)stream,
                       logits, logit_sl, logit_sq, boxes, box_sl, box_sq, gt_labels, gt_boxes, n_layers, Q, K, T,
                       w_class, w_bbox, w_giou, cost);
    return check_launch("match_cost_kernel");
}

int clipops_pair_box_loss_fwd_f32(const float *boxes, const int64_t *lay, const int64_t *qidx, long row_mul,
                                  long row_add, const float *tgt_boxes, const int64_t *gidx, const float *weight,
                                  int n, float *l1, float *giou_loss, void *stream) {
    if (n < 0) return fail(1, "clipops_pair_box_loss_fwd_f32: negative count");
    if (n == 0) { g_err[0] = 0; return 0; }
    if (!boxes || !lay || !qidx || !tgt_boxes || !l1 || !giou_loss)
        return fail(1, "clipops_pair_box_loss_fwd_f32: null pointer");
    hipLaunchKernelGGL(pair_box_loss_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, lay,
                       qidx, row_mul, row_add, tgt_boxes, gidx, weight, n, l1, giou_loss);
    return check_launch("pair_box_loss_fwd_kernel");
}

int clipops_pair_box_loss_bwd_f32(const float *boxes, const int64_t *lay, const int64_t *qidx, long row_mul,
                                  long row_add, const float *tgt_boxes, const int64_t *gidx, const float *weight,
                                  int n, const float *grad_l1, const float *grad_giou, float *grad_boxes,
                                  void *stream) {
    if (n < 0) return fail(1, "clipops_pair_box_loss_bwd_f32: negative count");
    if (n == 0) { g_err[0] = 0; return 0; }
    if (!boxes || !lay || !qidx || !tgt_boxes || !grad_l1 || !grad_giou || !grad_boxes)
        return fail(1, "clipops_pair_box_loss_bwd_f32: null pointer");
    hipLaunchKernelGGL(pair_box_loss_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, lay,
                       qidx, row_mul, row_add, tgt_boxes, gidx, weight, n, grad_l1, grad_giou, grad_boxes);
    return check_launch("pair_box_loss_bwd_kernel");
}

int clipops_track_ownership_i64(const int64_t *track_ids, int n_tracks, const int64_t *gt_ids, int n_gt,
                                int64_t *matched_idx, float *gt_free, void *stream) {
    if (n_tracks < 0 || n_gt < 0) return fail(1, "clipops_track_ownership_i64: negative count");
    if (n_tracks + n_gt == 0) { g_err[0] = 0; return 0; }
    if ((n_tracks > 0 && (!track_ids || !matched_idx)) || (n_gt > 0 && (!gt_ids || !gt_free)))
        return fail(1, "clipops_track_ownership_i64: null pointer");
    hipLaunchKernelGGL(track_ownership_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, track_ids, n_tracks, gt_ids,
                       n_gt, matched_idx, gt_free);
    return check_launch("track_ownership_kernel");
}

int clipops_focal_labels_i64(const int64_t *lay, const int64_t *q, const int64_t *g, int n_pairs,
                             const int64_t *gt_labels, int n_gt, const int64_t *matched_idx, int n_tracks,
                             const uint8_t *late, int n_layers, int n_det, int K, int64_t *labels, void *stream) {
    if (n_pairs < 0 || n_gt < 0 || n_tracks < 0 || n_layers < 0 || n_det < 0) return fail(1, "clipops_focal_labels_i64: negative count");
    if ((long)n_layers * (n_det + n_tracks) == 0) { g_err[0] = 0; return 0; }
    if (!labels || !late || (n_pairs > 0 && (!lay || !q || !g || !gt_labels)) || (n_tracks > 0 && !matched_idx) ||
        (n_gt > 0 && !gt_labels))
        return fail(1, "clipops_focal_labels_i64: null pointer");
    hipLaunchKernelGGL(focal_labels_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lay, q, g, n_pairs, gt_labels, n_gt,
                       matched_idx, n_tracks, late, n_layers, n_det, K, labels);
    return check_launch("focal_labels_kernel");
}

int clipops_pair_iou_f32(const float *boxes, const float *tgt_boxes, const int64_t *gidx, int n, float *iou,
                         void *stream) {
    if (n < 0) return fail(1, "clipops_pair_iou_f32: negative count");
    if (n == 0) { g_err[0] = 0; return 0; }
    if (!boxes || !tgt_boxes || !iou) return fail(1, "clipops_pair_iou_f32: null pointer");
    hipLaunchKernelGGL(pair_iou_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, tgt_boxes,
                       gidx, n, iou);
    return check_launch("pair_iou_kernel");
}

int clipops_focal_fwd_f32(const float *logits, long sl, long sq, const int64_t *labels, int n_layers, int Nq, int K,
                          float alpha, float gamma, float *loss, void *stream) {
    if (n_layers < 0 || Nq < 0 || K <= 0) return fail(1, "clipops_focal_fwd_f32: bad dimension");
    if (n_layers == 0) { g_err[0] = 0; return 0; }
    if (!loss || (Nq > 0 && (!logits || !labels))) return fail(1, "clipops_focal_fwd_f32: null pointer");
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(n_layers), dim3(256), 0, (hipStream_t)stream, logits, sl, sq, labels, Nq,
                       K, alpha, gamma, loss);
    return check_launch("focal_fwd_kernel");
}

int clipops_focal_bwd_f32(const float *logits, long sl, long sq, const int64_t *labels, int n_layers, int Nq, int K,
                          float alpha, float gamma, const float *grad_loss, float *grad_logits, void *stream) {
    if (n_layers < 0 || Nq < 0 || K <= 0) return fail(1, "clipops_focal_bwd_f32: bad dimension");
    if ((long)n_layers * Nq == 0) { g_err[0] = 0; return 0; }
    if (!logits || !labels || !grad_loss || !grad_logits) return fail(1, "clipops_focal_bwd_f32: null pointer");
    hipLaunchKernelGGL(focal_bwd_kernel, dim3(grid_for((long)n_layers * Nq * K)), dim3(256), 0, (hipStream_t)stream,
                       logits, sl, sq, labels, n_layers, Nq, K, alpha, gamma, grad_loss, grad_logits);
    return check_launch("focal_bwd_kernel");
}

int clipops_sine_embed_fwd_f32(const float *pos, const float *dim_t, long n, int K, int F, float scale, float *out,
                               void *stream) {
    if (n < 0 || K <= 0 || F <= 0) return fail(1, "clipops_sine_embed_fwd_f32: bad dimension");
    if (n == 0) { g_err[0] = 0; return 0; }
    if (!pos || !dim_t || !out) return fail(1, "clipops_sine_embed_fwd_f32: null pointer");
    hipLaunchKernelGGL(sine_embed_fwd_kernel, dim3(grid_for(n * K * F)), dim3(256), 0, (hipStream_t)stream, pos, dim_t,
                       n, K, F, scale, out);
    return check_launch("sine_embed_fwd_kernel");
}

int clipops_sine_embed_bwd_f32(const float *pos, const float *dim_t, long n, int K, int F, float scale,
                               const float *grad_out, float *grad_pos, void *stream) {
    if (n < 0 || K <= 0 || F <= 0) return fail(1, "clipops_sine_embed_bwd_f32: bad dimension");
    if (n == 0) { g_err[0] = 0; return 0; }
    if (!pos || !dim_t || !grad_out || !grad_pos) return fail(1, "clipops_sine_embed_bwd_f32: null pointer");
    hipLaunchKernelGGL(sine_embed_bwd_kernel, dim3(grid_for(n * K * 64)), dim3(256), 0, (hipStream_t)stream, pos,
                       dim_t, n, K, F, scale, grad_out, grad_pos);
    return check_launch("sine_embed_bwd_kernel");
}

int clipops_inverse_sigmoid_fwd_f32(const float *x, long n, float eps, float *y, void *stream) {
    if (n < 0) return fail(1, "clipops_inverse_sigmoid_fwd_f32: negative count");
    if (n == 0) { g_err[0] = 0; return 0; }
    if (!x || !y) return fail(1, "clipops_inverse_sigmoid_fwd_f32: null pointer");
    hipLaunchKernelGGL(inverse_sigmoid_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, eps, y);
    return check_launch("inverse_sigmoid_fwd_kernel");
}

int clipops_inverse_sigmoid_bwd_f32(const float *x, const float *grad_y, long n, float eps, float *grad_x,
                                    void *stream) {
    if (n < 0) return fail(1, "clipops_inverse_sigmoid_bwd_f32: negative count");
    if (n == 0) { g_err[0] = 0; return 0; }
    if (!x || !grad_y || !grad_x) return fail(1, "clipops_inverse_sigmoid_bwd_f32: null pointer");
    hipLaunchKernelGGL(inverse_sigmoid_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, grad_y, n,
                       eps, grad_x);
    return check_launch("inverse_sigmoid_bwd_kernel");
}

int clipops_refine_boxes_fwd_f32(const float *delta, const float *ref, long n, float eps, float *out, void *stream) {
    if (n < 0) return fail(1, "clipops_refine_boxes_fwd_f32: negative count");
    if (n == 0) { g_err[0] = 0; return 0; }
    if (!delta || !ref || !out) return fail(1, "clipops_refine_boxes_fwd_f32: null pointer");
    hipLaunchKernelGGL(refine_boxes_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, delta, ref, n,
                       eps, out);
    return check_launch("refine_boxes_fwd_kernel");
}

int clipops_refine_boxes_bwd_f32(const float *out, const float *ref, const float *grad_out, long n, float eps,
                                 float *grad_delta, float *grad_ref, void *stream) {
    if (n < 0) return fail(1, "clipops_refine_boxes_bwd_f32: negative count");
    if (n == 0) { g_err[0] = 0; return 0; }
    if (!out || !ref || !grad_out || !grad_delta) return fail(1, "clipops_refine_boxes_bwd_f32: null pointer");
    hipLaunchKernelGGL(refine_boxes_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, ref,
                       grad_out, n, eps, grad_delta, grad_ref);
    return check_launch("refine_boxes_bwd_kernel");
}

int clipops_colsum_f32(const float *x, long rows, int cols, float *out, void *stream) {
    if (rows < 0 || cols < 0) return fail(1, "clipops_colsum_f32: bad dimension");
    if (cols == 0) { g_err[0] = 0; return 0; }
    if (!out || (rows > 0 && !x)) return fail(1, "clipops_colsum_f32: null pointer");
    if (cols % 4 == 0 && rows <= 2048 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(colsum_quad_kernel, dim3(cols / 4), dim3(256), 0, (hipStream_t)stream, x, rows, cols, out);
        return check_launch("colsum_quad_kernel");
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, rows, cols, out);
    return check_launch("colsum_kernel");
}

int clipops_colsum_partial_f32(const float *x, long rows, int cols, int chunk_rows, float *partial, void *stream) {
    if (rows < 0 || cols < 0 || chunk_rows <= 0) return fail(1, "clipops_colsum_partial_f32: bad dimension");
    if (cols == 0 || rows == 0) { g_err[0] = 0; return 0; }
    if (!x || !partial) return fail(1, "clipops_colsum_partial_f32: null pointer");
    const long chunks = (rows + chunk_rows - 1) / chunk_rows;
    if (chunks > 65535) return fail(1, "clipops_colsum_partial_f32: too many chunks");
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3((cols + 31) / 32, (unsigned)chunks), dim3(256), 0,
                       (hipStream_t)stream, x, rows, cols, chunk_rows, partial);
    return check_launch("colsum_partial_kernel");
}

int clipops_colsum_partial_bf16(const uint16_t *x, long rows, int cols, int chunk_rows, float *partial, void *stream) {
    if (rows < 0 || cols < 0 || chunk_rows <= 0) return fail(1, "clipops_colsum_partial_bf16: bad dimension");
    if (cols == 0 || rows == 0) { g_err[0] = 0; return 0; }
    if (!x || !partial) return fail(1, "clipops_colsum_partial_bf16: null pointer");
    const long chunks = (rows + chunk_rows - 1) / chunk_rows;
    if (chunks > 65535) return fail(1, "clipops_colsum_partial_bf16: too many chunks");
    hipLaunchKernelGGL(colsum_partial_kernel<uint16_t>, dim3((cols + 31) / 32, (unsigned)chunks), dim3(256), 0,
                       (hipStream_t)stream, x, rows, cols, chunk_rows, partial);
    return check_launch("colsum_partial_kernel<bf16>");
}

static int relu_bwd_colsum_check(const void *g, const void *y, const void *g2, const void *partial, long rows, int cols,
                                 int chunk_rows, long *chunks) {
    if (rows < 0 || cols < 0 || chunk_rows <= 0) return fail(1, "clipops_relu_bwd_colsum_partial: bad dimension");
    *chunks = (rows + chunk_rows - 1) / chunk_rows;
    if (cols == 0 || rows == 0) return 0;
    if (!g || !y || !g2 || !partial) return fail(1, "clipops_relu_bwd_colsum_partial: null pointer");
    if (*chunks > 65535) return fail(1, "clipops_relu_bwd_colsum_partial: too many chunks");
    return 0;
}

int clipops_relu_bwd_colsum_partial_f32(const float *g, const float *y, long rows, int cols, int chunk_rows, float *g2,
                                        float *partial, void *stream) {
    long chunks = 0;
    if (relu_bwd_colsum_check(g, y, g2, partial, rows, cols, chunk_rows, &chunks)) return 1;
    if (cols == 0 || rows == 0) { g_err[0] = 0; return 0; }
    hipLaunchKernelGGL(relu_bwd_colsum_partial_kernel<float>, dim3((cols + 31) / 32, (unsigned)chunks), dim3(256), 0,
                       (hipStream_t)stream, g, y, rows, cols, chunk_rows, g2, partial);
    return check_launch("relu_bwd_colsum_partial_kernel");
}

int clipops_mha_fwd_f32(const float *q, const float *k, const float *v, long q_bs, long q_rs, long k_bs, long k_rs,
                        long v_bs, long v_rs, const uint8_t *key_mask, int B, int H, int L, float scale, float *out,
                        float *lse, void *stream) {
    int rc = mha_check(q, k, v, B, H, L);
    if (rc) return rc;
    if ((long)B * L == 0) { g_err[0] = 0; return 0; }
    if (!out || !lse) return fail(1, "clipops_mha_fwd_f32: null pointer");
    const size_t lds = (size_t)2 * L * kPitch * sizeof(float);
    if ((rc = allow_lds(reinterpret_cast<const void *>(mha_fwd_kernel), lds, g_lds_fwd))) return rc;
    hipLaunchKernelGGL(mha_fwd_kernel, dim3((L + kRows - 1) / kRows, H, B), dim3(256), lds, (hipStream_t)stream, q, k, v, q_bs,
                       q_rs, k_bs, k_rs, v_bs, v_rs, key_mask, H, L, scale, out, lse);
    return check_launch("mha_fwd_kernel");
}

int clipops_mha_bwd_f32(const float *q, const float *k, const float *v, long q_bs, long q_rs, long k_bs, long k_rs,
                        long v_bs, long v_rs, const uint8_t *key_mask, const float *out, const float *lse,
                        const float *grad_out, int B, int H, int L, float scale, float *grad_q, long gq_bs, long gq_rs,
                        float *grad_k, long gk_bs, long gk_rs, float *grad_v, long gv_bs, long gv_rs, void *stream) {
    int rc = mha_check(q, k, v, B, H, L);
    if (rc) return rc;
    if ((long)B * L == 0) { g_err[0] = 0; return 0; }
    if (!out || !lse || !grad_out || !grad_q || !grad_k || !grad_v) return fail(1, "clipops_mha_bwd_f32: null pointer");
    const size_t lds_q = (size_t)2 * L * kPitch * sizeof(float);
    const size_t lds_kv = lds_q + (size_t)2 * L * sizeof(float);
    if ((rc = allow_lds(reinterpret_cast<const void *>(mha_bwd_q_kernel), lds_q, g_lds_bq))) return rc;
    if ((rc = allow_lds(reinterpret_cast<const void *>(mha_bwd_kv_kernel), lds_kv, g_lds_bkv))) return rc;
    const dim3 grid((L + kRows - 1) / kRows, H, B);
    hipLaunchKernelGGL(mha_bwd_q_kernel, grid, dim3(256), lds_q, (hipStream_t)stream, q, k, v, q_bs, q_rs, k_bs, k_rs,
                       v_bs, v_rs, key_mask, out, lse, grad_out, H, L, scale, grad_q, gq_bs, gq_rs);
    if ((rc = check_launch("mha_bwd_q_kernel"))) return rc;
    hipLaunchKernelGGL(mha_bwd_kv_kernel, grid, dim3(256), lds_kv, (hipStream_t)stream, q, k, v, q_bs, q_rs, k_bs, k_rs,
                       v_bs, v_rs, key_mask, out, lse, grad_out, H, L, scale, grad_k, gk_bs, gk_rs, grad_v, gv_bs, gv_rs);
    return check_launch("mha_bwd_kv_kernel");
}

int clipops_add_layer_norm_fwd_f32(const float *x, const float *res, const float *gamma, const float *beta, long rows,
                                   float eps, float *sum, float *y, float *stats, void *stream) {
    if (rows < 0) return fail(1, "clipops_add_layer_norm_fwd_f32: negative row count");
    if (rows == 0) { g_err[0] = 0; return 0; }
    if (!x || !res || !gamma || !beta || !sum || !y || !stats) return fail(1, "clipops_add_layer_norm_fwd_f32: null pointer");
    hipLaunchKernelGGL(add_layer_norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                       res, gamma, beta, rows, eps, sum, y, stats);
    return check_launch("add_layer_norm_fwd_kernel");
}

int clipops_add_layer_norm_bwd_f32(const float *grad_y, const float *sum, const float *stats, const float *gamma,
                                   long rows, int chunk_rows, float *grad_sum, float *partial, void *stream) {
    if (rows < 0 || chunk_rows <= 0) return fail(1, "clipops_add_layer_norm_bwd_f32: bad dimension");
    if (rows == 0) { g_err[0] = 0; return 0; }
    if (!grad_y || !sum || !stats || !gamma || !grad_sum || !partial)
        return fail(1, "clipops_add_layer_norm_bwd_f32: null pointer");
    hipLaunchKernelGGL(add_layer_norm_bwd_kernel, dim3((unsigned)((rows + chunk_rows - 1) / chunk_rows)), dim3(256), 0,
                       (hipStream_t)stream, grad_y, sum, stats, gamma, rows, chunk_rows, grad_sum, partial);
    return check_launch("add_layer_norm_bwd_kernel");
}

int clipops_assign_f32(const float *cost, long stride_problem, long stride_row, long stride_col, int n_problems,
                       int n_rows, int n_cols, int32_t *row_ind, int32_t *col_ind, int32_t *status, void *stream) {
    if (n_problems < 0 || n_rows < 0 || n_cols < 0) return fail(1, "clipops_assign_f32: bad dimension");
    if (n_problems == 0 || n_rows == 0 || n_cols == 0) { g_err[0] = 0; return 0; }
    if (!cost || !row_ind || !col_ind) return fail(1, "clipops_assign_f32: null pointer");
    const int nr = n_rows < n_cols ? n_rows : n_cols, nc = n_rows < n_cols ? n_cols : n_rows;
    if (nc > CLIPOPS_ASSIGN_MAX_DIM) return fail(2, "clipops_assign_f32: problem exceeds CLIPOPS_ASSIGN_MAX_DIM");
    const size_t lds = (assign::work_bytes(nr, nc) + 15) & ~(size_t)15;
    if (lds > 64 * 1024) {       // beyond the default limit of a launch: opt in (gfx950 has 160 KB per CU), once per process
        static std::atomic<size_t> allowed{0};
        if (lds > 160 * 1024) return fail(2, "clipops_assign_f32: problem does not fit the LDS of a CU");
        if (lds > allowed.load()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(assign_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                return fail(3, "clipops_assign_f32: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
            }
            allowed.store(160 * 1024);
        }
    }
    hipLaunchKernelGGL(assign_kernel, dim3(n_problems), dim3(64), lds, (hipStream_t)stream, cost, stride_problem,
                       stride_row, stride_col, n_rows, n_cols, row_ind, col_ind, status);
    return check_launch("assign_kernel");
}

static int shift_relu_check(const void *x, const void *shift, long planes, int C, long HW, const char *who) {
    if (planes < 0 || C <= 0 || HW < 0) return fail(1, who);
    if (planes && HW && (!x || !shift)) return fail(1, who);
    return 0;
}

int clipops_shift_relu_f32(float *x, const float *shift, const float *res, long planes, int C, long HW, void *stream) {
    if (shift_relu_check(x, shift, planes, C, HW, "clipops_shift_relu_f32: bad argument")) return 1;
    if (planes == 0 || HW == 0) { g_err[0] = 0; return 0; }
    const int vec = (HW % 4 == 0) && ((uintptr_t)x % 16 == 0) && (!res || (uintptr_t)res % 16 == 0);
    const long per = vec ? HW / 4 : HW;
    const dim3 grid((unsigned)((per + 1023) / 1024 > 64 ? 64 : (per + 1023) / 1024 < 1 ? 1 : (per + 1023) / 1024),
                    (unsigned)(planes > 65535 ? 65535 : planes));
    if (res) hipLaunchKernelGGL(shift_relu_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, shift, res, planes, C, HW, vec);
    else hipLaunchKernelGGL(shift_relu_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, shift, res, planes, C, HW, vec);
    return check_launch("shift_relu_f32_kernel");
}

int clipops_shift_relu_bf16(uint16_t *x, const float *shift, const uint16_t *res, long planes, int C, long HW, void *stream) {
    if (shift_relu_check(x, shift, planes, C, HW, "clipops_shift_relu_bf16: bad argument")) return 1;
    if (planes == 0 || HW == 0) { g_err[0] = 0; return 0; }
    const int vec = (HW % 8 == 0) && ((uintptr_t)x % 16 == 0) && (!res || (uintptr_t)res % 16 == 0);
    const long per = vec ? HW / 8 : HW;
    const dim3 grid((unsigned)((per + 1023) / 1024 > 64 ? 64 : (per + 1023) / 1024 < 1 ? 1 : (per + 1023) / 1024),
                    (unsigned)(planes > 65535 ? 65535 : planes));
    if (res) hipLaunchKernelGGL(shift_relu_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, shift, res, planes, C, HW, vec);
    else hipLaunchKernelGGL(shift_relu_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, shift, res, planes, C, HW, vec);
    return check_launch("shift_relu_bf16_kernel");
}

int clipops_linear_bwd_f32(const float *grad_y, const float *y_relu, const float *x, const float *w, int rows,
                           int in_features, int out_features, float *grad_x, float *grad_w, float *grad_b,
                           void *stream) {
    if (rows < 0 || in_features <= 0 || out_features <= 0) return fail(1, "clipops_linear_bwd_f32: bad dimension");
    if (!grad_y || !x || !w) return fail(1, "clipops_linear_bwd_f32: null pointer");
    if (grad_b && !grad_w) return fail(1, "clipops_linear_bwd_f32: grad_b is computed with grad_w");
    const int gx_ti = (rows + 31) / 32, gx_tj = (in_features + 31) / 32;
    const int gw_ti = (out_features + 31) / 32, gw_tj = gx_tj;
    const int n_gx = grad_x ? gx_ti * gx_tj : 0, n_gw = grad_w ? gw_ti * gw_tj : 0;
    if (n_gx + n_gw == 0) { g_err[0] = 0; return 0; }
    if (rows == 0) {          // nothing to contract over: the weight and bias gradients are zero
        if (grad_w && hipMemsetAsync(grad_w, 0, sizeof(float) * (size_t)out_features * in_features, (hipStream_t)stream) != hipSuccess)
            return fail(3, "clipops_linear_bwd_f32: memset failed");
        if (grad_b && hipMemsetAsync(grad_b, 0, sizeof(float) * (size_t)out_features, (hipStream_t)stream) != hipSuccess)
            return fail(3, "clipops_linear_bwd_f32: memset failed");
        g_err[0] = 0;
        return 0;
    }
    if (y_relu) hipLaunchKernelGGL(linear_bwd_kernel<true>, dim3(n_gx + n_gw), dim3(256), 0, (hipStream_t)stream, grad_y, y_relu, x, w, rows, in_features, out_features, grad_x, grad_w, grad_b, n_gx, gx_tj, gw_tj);
    else hipLaunchKernelGGL(linear_bwd_kernel<false>, dim3(n_gx + n_gw), dim3(256), 0, (hipStream_t)stream, grad_y, y_relu, x, w, rows, in_features, out_features, grad_x, grad_w, grad_b, n_gx, gx_tj, gw_tj);
    return check_launch("linear_bwd_kernel");
}

int clipops_linear_fwd_f32(const float *x, const float *w, const float *bias, int rows, int in_features,
                           int out_features, int relu, float *y, void *stream) {
    if (rows < 0 || in_features <= 0 || out_features <= 0) return fail(1, "clipops_linear_fwd_f32: bad dimension");
    if (in_features % 4) return fail(2, "clipops_linear_fwd_f32: in_features must be a multiple of 4");
    if (rows == 0) { g_err[0] = 0; return 0; }
    if (!x || !w || !y) return fail(1, "clipops_linear_fwd_f32: null pointer");
    const int ti = (rows + 31) / 32, tj = (out_features + 31) / 32;
    if (relu) hipLaunchKernelGGL(linear_fwd_kernel<true>, dim3(ti * tj), dim3(256), 0, (hipStream_t)stream, x, w, bias, rows, in_features, out_features, y, tj);
    else hipLaunchKernelGGL(linear_fwd_kernel<false>, dim3(ti * tj), dim3(256), 0, (hipStream_t)stream, x, w, bias, rows, in_features, out_features, y, tj);
    return check_launch("linear_fwd_kernel");
}

}  // extern "C"
