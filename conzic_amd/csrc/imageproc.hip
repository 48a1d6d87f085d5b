// CLIP image pre-processing on the device (clip/clip.py:55-56 -> HF CLIPImageProcessor):
// RGB uint8 [H][W][3] -> bicubic resize of the shorter side to S -> centre crop SxS -> /255 -> normalise -> CHW fp32.
//
// The resize restates Pillow's 8-bit resampler (third-party dependency of the reference, PIL `Image.resize(size,
// BICUBIC)` under transformers' PIL image backend; algorithm of libImaging/Resample.c): per output coordinate a
// window [xmin, xmin+n) of the input with weights w(x) = cubic((x + xmin - centre + 0.5) / max(scale,1)), a = -0.5,
// support 2*max(scale,1), normalised to sum 1 in double, then rounded to 22-bit fixed point; a pass accumulates
// 2^21 + sum(pixel * coeff) in int32, shifts by 22 and clamps to 0..255.  Horizontal pass first, then vertical,
// each through a uint8 intermediate, and a pass is skipped when that extent does not change -- all of which is
// reproduced here so that the result is bit-identical to the host path (tests/test_imageproc_gpu.py).
#include <cmath>
#include <vector>

#include "kernels.h"

namespace czc {

namespace {

constexpr int PREC_BITS = 32 - 8 - 2;

inline double cubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// bounds[2*i] = first input index, bounds[2*i+1] = taps; coeff[i*ksize + t] fixed point
int precompute(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& coeff) {
  const double scale = (double)in_size / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  coeff.assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = cubic((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      const double v = k[x];
      coeff[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PREC_BITS)) : (int)(0.5 + v * (1 << PREC_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return ksize;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PREC_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// out[y][xx][c] = clip8(2^21 + sum_t in[y][xmin+t][c] * k[xx][t]);  one thread per output pixel
__global__ void resample_h_kernel(const unsigned char* in, int H, int W, unsigned char* out, int OW, const int* bounds,
                                  const int* coeff, int ksize) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (xx >= OW) return;
  const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = coeff + (size_t)xx * ksize;
  int s0 = 1 << (PREC_BITS - 1), s1 = s0, s2 = s0;
  const unsigned char* p = in + ((size_t)y * W + xmin) * 3;
  for (int t = 0; t < n; ++t) {
    const int c = k[t];
    s0 += p[3 * t] * c; s1 += p[3 * t + 1] * c; s2 += p[3 * t + 2] * c;
  }
  unsigned char* o = out + ((size_t)y * OW + xx) * 3;
  o[0] = (unsigned char)clip8(s0); o[1] = (unsigned char)clip8(s1); o[2] = (unsigned char)clip8(s2);
}

// vertical pass (or none) restricted to the crop window, fused with /255, normalise and HWC -> CHW
__global__ void resample_v_crop_norm_kernel(const unsigned char* in, int IH, int IW, int vertical, const int* bounds,
                                            const int* coeff, int ksize, int top, int left, int S, float m0, float m1,
                                            float m2, float d0, float d1, float d2, float* out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= S) return;
  const int sx = x + left, oy = y + top;
  int v0, v1, v2;
  if (vertical) {
    const int ymin = bounds[2 * oy], n = bounds[2 * oy + 1];
    const int* k = coeff + (size_t)oy * ksize;
    int s0 = 1 << (PREC_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
      const unsigned char* p = in + ((size_t)(ymin + t) * IW + sx) * 3;
      const int c = k[t];
      s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
    }
    v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
  } else {
    const unsigned char* p = in + ((size_t)oy * IW + sx) * 3;
    v0 = p[0]; v1 = p[1]; v2 = p[2];
  }
  // HF image processor in fp32: (v / 255 - mean) / std -- the same three IEEE operations (pinned bit-exact by
  // tests/golden/imageproc_*.npz, generated through the reference's CLIPProcessor)
  const size_t o = (size_t)y * S + x, plane = (size_t)S * S;
  out[o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.0f), m0), d0);
  out[plane + o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.0f), m1), d1);
  out[2 * plane + o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.0f), m2), d2);
}

}  // namespace

// rgb_dev: uint8 [H][W][3] on the device; scratch: >= H*NW*3 bytes + room for the tables (see imageproc_scratch_bytes)
size_t imageproc_scratch_bytes(int H, int W, int S) {
  const int nw = W <= H ? S : (int)((double)S * W / H), nh = W <= H ? (int)((double)S * H / W) : S;
  const size_t tables = ((size_t)nw + nh) * (2 + 2 * ((size_t)std::ceil(2.0 * std::max(1.0, std::max((double)W / nw, (double)H / nh))) * 2 + 1)) * 4;
  return (size_t)H * nw * 3 + 256 + tables + 4096;
}

int launch_clip_preprocess(const unsigned char* rgb_dev, int H, int W, int S, const float* mean, const float* stdv,
                           unsigned char* scratch, float* out, hipStream_t st) {
  if (H <= 0 || W <= 0 || S <= 0) {
    snprintf(g_err, sizeof(g_err), "preprocess: bad image size %dx%d", W, H);
    return 1;
  }
  // HF get_resize_output_image_size(shortest_edge=S, default_to_square=False): new_long = int(S * long / short)
  int nw, nh;
  if (W <= H) { nw = S; nh = (int)((double)S * H / W); }
  else { nw = (int)((double)S * W / H); nh = S; }
  if ((W == S && H == S)) { nw = W; nh = H; }
  const bool need_h = nw != W, need_v = nh != H;
  const int left = (nw - S) / 2, top = (nh - S) / 2;
  std::vector<int> bh, ch, bv, cv;
  int kh = 0, kv = 0;
  if (need_h) kh = precompute(W, nw, bh, ch);
  if (need_v) kv = precompute(H, nh, bv, cv);
  unsigned char* tmp = scratch;  // [H][nw][3] when the horizontal pass runs
  size_t off = ((size_t)H * nw * 3 + 255) & ~(size_t)255;
  int* d_bh = (int*)(scratch + off); off += bh.size() * 4;
  int* d_ch = (int*)(scratch + off); off += ch.size() * 4;
  int* d_bv = (int*)(scratch + off); off += bv.size() * 4;
  int* d_cv = (int*)(scratch + off); off += cv.size() * 4;
  if (need_h) {
    CZC_HIP_CHECK(hipMemcpyAsync(d_bh, bh.data(), bh.size() * 4, hipMemcpyHostToDevice, st));
    CZC_HIP_CHECK(hipMemcpyAsync(d_ch, ch.data(), ch.size() * 4, hipMemcpyHostToDevice, st));
  }
  if (need_v) {
    CZC_HIP_CHECK(hipMemcpyAsync(d_bv, bv.data(), bv.size() * 4, hipMemcpyHostToDevice, st));
    CZC_HIP_CHECK(hipMemcpyAsync(d_cv, cv.data(), cv.size() * 4, hipMemcpyHostToDevice, st));
  }
  // the staging vectors die with this frame: the copies must have left them
  CZC_HIP_CHECK(hipStreamSynchronize(st));
  const unsigned char* src = rgb_dev;
  int IW = W;
  if (need_h) {
    hipLaunchKernelGGL(resample_h_kernel, dim3(cdiv(nw, 128), H), dim3(128), 0, st, rgb_dev, H, W, tmp, nw, d_bh, d_ch, kh);
    src = tmp;
    IW = nw;
  }
  hipLaunchKernelGGL(resample_v_crop_norm_kernel, dim3(cdiv(S, 128), S), dim3(128), 0, st, src, H, IW, need_v ? 1 : 0, d_bv,
                     d_cv, kv, top, left, S, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], out);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc
