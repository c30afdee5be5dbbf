// evaluator.hip -- device side of the multi-scale / flip sliding-window evaluator, the direct caller of the forward
// (additional_utils/encoding_models.py:54-155 MultiEvalModule.forward / module_inference / pad_image / crop_image / flip_image,
//  additional_utils/models.py:55-140 LSeg_MultiEvalModule.forward).  The reference does this with one torch op
// per crop (pad per channel, slice, flip, slice-add, count) around 2 x n_crops B = 1 forwards; here every scale is
//   make_crops  : resized image -> the whole stack of padded crops AND their mirrored twins, one pass
//   (the engine runs the stack as one batch)
//   accumulate  : the stack of logits -> count-normalised score map of the scale (flip-add, overlap-add in the reference's box order,
//                 divide, crop to the un-padded size), one pass, no atomics
//   resize_add  : scores (+)= bilinear(map) at the input resolution (also used for the image resize itself)
#include "ops.h"
#include "../../include/lseg_hip.h"

namespace lseg {
namespace {

// crops [(1 + flip) * n, C, crop, crop]; crop k = (idh, idw) starts at (idh * stride, idw * stride) of the (virtually padded) image;
// pixels outside [0, height) x [0, width) take pad[c] = -mean[c] / std[c] (pad_image, :144-155); twin n + k = crop k mirrored in x
__global__ void eval_make_crops_kernel(const float* __restrict__ img, float* __restrict__ crops, int C, int height, int width, int crop,
                                       int stride, int w_grids, int n, int flip, float pad0, float pad1, float pad2) {
    const size_t total = (size_t)(flip ? 2 : 1) * n * C * crop * crop;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % crop);
        size_t p = i / crop;
        const int y = (int)(p % crop); p /= crop;
        const int c = (int)(p % C);
        const int kf = (int)(p / C);
        const int k = kf >= n ? kf - n : kf;
        const int xs = kf >= n ? crop - 1 - x : x;
        const int idh = k / w_grids, idw = k - idh * w_grids;
        const int yy = idh * stride + y, xx = idw * stride + xs;
        const float pv = c == 0 ? pad0 : (c == 1 ? pad1 : pad2);
        crops[i] = (yy < height && xx < width) ? img[((size_t)c * height + yy) * width + xx] : pv;
    }
}

// outputs [K, height, width]: sum over the boxes covering the pixel, in the reference's (idh, idw) order, of
// outs[k] + flip_x(outs[n + k]), divided by the number of covering boxes (encoding_models.py:100-121)
__global__ void eval_accumulate_kernel(const float* __restrict__ outs, float* __restrict__ outputs, int K, int height, int width, int ph, int pw,
                                       int crop, int stride, int h_grids, int w_grids, int flip) {
    const int n = h_grids * w_grids;
    const size_t total = (size_t)K * height * width;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % width);
        const int y = (int)((i / width) % height);
        const int k = (int)(i / ((size_t)width * height));
        float acc = 0.f, cnt = 0.f;
        for (int idh = 0; idh < h_grids; ++idh) {
            const int h0 = idh * stride, h1 = min(h0 + crop, ph);
            if (y < h0 || y >= h1) continue;
            for (int idw = 0; idw < w_grids; ++idw) {
                const int w0 = idw * stride, w1 = min(w0 + crop, pw);
                if (x < w0 || x >= w1) continue;
                const int b = idh * w_grids + idw;
                const size_t o = (((size_t)b * K + k) * crop + (y - h0)) * crop;
                float v = outs[o + (x - w0)];
                if (flip) v += outs[o + (size_t)n * K * crop * crop + (crop - 1 - (x - w0))];
                acc += v;
                cnt += 1.f;
            }
        }
        outputs[i] = acc / cnt;
    }
}

// dst [P, Ho, Wo] (+)= bilinear(src [P, Hi, Wi]), align_corners=True (F.interpolate / upsample_bilinear2d's arithmetic)
__global__ void eval_resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int P, int Hi, int Wi, int Ho, int Wo, int accumulate) {
    const size_t total = (size_t)P * Ho * Wo;
    const float ry = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo);
        const int yo = (int)((i / Wo) % Ho);
        const size_t p = i / ((size_t)Wo * Ho);
        int y0, y1, x0, x1;
        float ly, lx;
        src_tap(ry, yo, Hi, y0, y1, ly);
        src_tap(rx, xo, Wi, x0, x1, lx);
        const float* s = src + p * (size_t)Hi * Wi;
        const float v = bilerp(s[(size_t)y0 * Wi + x0], s[(size_t)y0 * Wi + x1], s[(size_t)y1 * Wi + x0], s[(size_t)y1 * Wi + x1], lx, ly);
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

inline int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

int launch_eval_make_crops(const float* img, float* crops, int C, int height, int width, int crop, int stride, int h_grids, int w_grids,
                           int flip, const float* pad3, hipStream_t st) {
    const int n = h_grids * w_grids;
    hipLaunchKernelGGL(eval_make_crops_kernel, dim3(grid_for((size_t)(flip ? 2 : 1) * n * C * crop * crop)), dim3(256), 0, st, img, crops, C,
                       height, width, crop, stride, w_grids, n, flip, pad3[0], pad3[1], pad3[2]);
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_eval_accumulate(const float* outs, float* outputs, int K, int height, int width, int ph, int pw, int crop, int stride, int h_grids,
                           int w_grids, int flip, hipStream_t st) {
    hipLaunchKernelGGL(eval_accumulate_kernel, dim3(grid_for((size_t)K * height * width)), dim3(256), 0, st, outs, outputs, K, height, width,
                       ph, pw, crop, stride, h_grids, w_grids, flip);
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_eval_resize(const float* src, float* dst, int P, int Hi, int Wi, int Ho, int Wo, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(eval_resize_kernel, dim3(grid_for((size_t)P * Ho * Wo)), dim3(256), 0, st, src, dst, P, Hi, Wi, Ho, Wo, accumulate);
    LSEG_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace lseg
