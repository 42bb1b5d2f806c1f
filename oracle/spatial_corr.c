/*
 * ORACLE (test infrastructure, not product code): plain-C restatement of the
 * reference's CPU spatial correlation sampler,
 *   models/PWCNet/cpu_spatial_correlation_sampler-0.3.0/Correlation_Module/correlation.cpp
 *     correlate_patch        :9-37     correlate_patch_grad     :39-73
 *     correlation_cpp_forward:75-124   correlation_cpp_backward :126-178
 * Same loop nest, same accumulation order (c, then kernel rows i, then kernel
 * cols j; the backward accumulates over (ph, pw, h, w) serially per batch item),
 * so fp32 results are bit-identical to the reference build in oracle/_ref.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may call this.
 */
#include <string.h>

#define IDX3(c, y, x, H, W) (((long)(c) * (H) + (y)) * (W) + (x))

static int out_dim(int in, int pad, int k, int dil, int stride) {
  int dk = (k - 1) * dil + 1;
  return (in + 2 * pad - dk) / stride + 1;
}

int oracle_scorr_out_size(int iH, int iW, int kH, int kW, int padH, int padW, int dilH, int dilW,
                          int dH, int dW, int* oH, int* oW) {
  *oH = out_dim(iH, padH, kH, dilH, dH);
  *oW = out_dim(iW, padW, kW, dilW, dW);
  return 0;
}

/* correlation.cpp:9-37 */
static void correlate_patch(const float* in1, const float* in2, float* dst, int C, int iH, int iW,
                            int kH, int kW, int dilH, int dilW, int u, int v, int shiftU,
                            int shiftV) {
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < kH; ++i) {
      int i1 = u + i * dilH, i2 = i1 + shiftU;
      if (i1 >= 0 && i1 < iH && i2 >= 0 && i2 < iH)
        for (int j = 0; j < kW; ++j) {
          int j1 = v + j * dilW, j2 = j1 + shiftV;
          if (j1 >= 0 && j1 < iW && j2 >= 0 && j2 < iW)
            *dst += in1[IDX3(c, i1, j1, iH, iW)] * in2[IDX3(c, i2, j2, iH, iW)];
        }
    }
}

/* correlation.cpp:39-73 */
static void correlate_patch_grad(const float* in1, float* g1, const float* in2, float* g2, float go,
                                 int C, int iH, int iW, int kH, int kW, int dilH, int dilW, int u,
                                 int v, int shiftU, int shiftV) {
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < kH; ++i) {
      int i1 = u + i * dilH, i2 = i1 + shiftU;
      if (i1 >= 0 && i1 < iH && i2 >= 0 && i2 < iH)
        for (int j = 0; j < kW; ++j) {
          int j1 = v + j * dilW, j2 = j1 + shiftV;
          if (j1 >= 0 && j1 < iW && j2 >= 0 && j2 < iW) {
            float v1 = in1[IDX3(c, i1, j1, iH, iW)], v2 = in2[IDX3(c, i2, j2, iH, iW)];
            g2[IDX3(c, i2, j2, iH, iW)] += go * v1;
            g1[IDX3(c, i1, j1, iH, iW)] += go * v2;
          }
        }
    }
}

/* correlation.cpp:75-124 ; out [B][patchH][patchW][oH][oW] */
int oracle_scorr_forward(const float* in1, const float* in2, float* out, int B, int C, int iH, int iW,
                         int kH, int kW, int patchH, int patchW, int padH, int padW, int dilH,
                         int dilW, int dpH, int dpW, int dH, int dW) {
  const int prH = (patchH - 1) / 2, prW = (patchW - 1) / 2;
  const int oH = out_dim(iH, padH, kH, dilH, dH), oW = out_dim(iW, padW, kW, dilW, dW);
  const long plane = (long)C * iH * iW;
  memset(out, 0, sizeof(float) * (long)B * patchH * patchW * oH * oW);
#pragma omp parallel for collapse(2)
  for (int n = 0; n < B; ++n)
    for (int ph = 0; ph < patchH; ++ph)
      for (int pw = 0; pw < patchW; ++pw)
        for (int h = 0; h < oH; ++h)
          for (int w = 0; w < oW; ++w)
            correlate_patch(in1 + n * plane, in2 + n * plane,
                            out + ((((long)n * patchH + ph) * patchW + pw) * oH + h) * oW + w, C, iH,
                            iW, kH, kW, dilH, dilW, -padH + h * dH, -padW + w * dW,
                            (ph - prH) * dpH, (pw - prW) * dpW);
  return 0;
}

/* correlation.cpp:126-178 */
int oracle_scorr_backward(const float* in1, const float* in2, const float* gout, float* g1, float* g2,
                          int B, int C, int iH, int iW, int kH, int kW, int patchH, int patchW,
                          int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH, int dW) {
  const int prH = (patchH - 1) / 2, prW = (patchW - 1) / 2;
  const int oH = out_dim(iH, padH, kH, dilH, dH), oW = out_dim(iW, padW, kW, dilW, dW);
  const long plane = (long)C * iH * iW;
  memset(g1, 0, sizeof(float) * B * plane);
  memset(g2, 0, sizeof(float) * B * plane);
#pragma omp parallel for
  for (int n = 0; n < B; ++n)
    for (int ph = 0; ph < patchH; ++ph)
      for (int pw = 0; pw < patchW; ++pw)
        for (int h = 0; h < oH; ++h)
          for (int w = 0; w < oW; ++w)
            correlate_patch_grad(in1 + n * plane, g1 + n * plane, in2 + n * plane, g2 + n * plane,
                                 gout[((((long)n * patchH + ph) * patchW + pw) * oH + h) * oW + w], C,
                                 iH, iW, kH, kW, dilH, dilW, -padH + h * dH, -padW + w * dW,
                                 (ph - prH) * dpH, (pw - prW) * dpW);
  return 0;
}
