/*
 * pbd_oracle.c — CPU restatement of PartsBasedDetector<float>::detect().
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under partsbaseddetector_amd/ may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (wg-perception/PartsBasedDetector) ships no
 * test, golden vector or fixture for this path (test/CMakeLists.txt:1-10 only
 * registers ecto config tests) and cannot be built in this image (every hot
 * path TU needs OpenCV, which is absent; building it against stand-in headers
 * is not allowed).  Each function below therefore follows the reference
 * source line by line and cites it; cv::resize / cv::pyrDown / FilterEngine
 * (third-party OpenCV, version not pinned by the reference: package.xml:18,
 * .travis.yml:5-6) are restated from the published OpenCV 2.4 algorithms.
 *
 * Arithmetic notes: compile with -ffp-contract=off (the reference is built
 * for baseline x86-64: no FMA).  T = float throughout (src/demo.cpp:85).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/pbd_c.h"

#define ORC_API __attribute__((visibility("default")))

static inline int cv_round_f(float v) { return (int)lrint((double)v); } /* cvRound: half to even */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------------- */
/* Pyramid geometry: src/HOGFeatures.cpp:98-127, include/HOGFeatures.hpp:74-81 */
/* ------------------------------------------------------------------------- */
static float orc_sfactor(int interval) {
  /* sfactor_ = pow(2.0f, 1.0f/(float)interval_)  (HOGFeatures.hpp:78) */
  return (float)pow(2.0, (double)(1.0f / (float)interval));
}

ORC_API int orc_cells_of(int iw, int ih, int sbin, int* cw, int* ch) {
  /* src/HOGFeatures.cpp:174-175: blocks = round(size / sbin), out = blocks - 2 */
  int bw = (int)roundf((float)iw / (float)sbin);
  int bh = (int)roundf((float)ih / (float)sbin);
  *cw = imax(bw - 2, 0);
  *ch = imax(bh - 2, 0);
  return 0;
}

ORC_API int orc_pyramid_geometry(int w, int h, int sbin, int interval, int* nlevels,
                                 int32_t* iw, int32_t* ih, int32_t* cw, int32_t* ch, float* scales) {
  float sf = orc_sfactor(interval);
  float fw = (float)w, fh = (float)h;
  /* :99  nscales_ = 1 + floor(log(min(h,w)/(5.0f*sbin))/log(sfactor_)) in float */
  float mn = fh < fw ? fh : fw;
  float r = logf(mn / (5.0f * (float)sbin)) / logf(sf);
  int n = (int)(1.0f + floorf(r));
  if (n < interval) return -1; /* reference writes out of range here */
  *nlevels = n;
  if (!iw) return 0;
  for (int i = 0; i < interval; ++i) {
    /* :116 imsize * (float)(1.0f/pow(sfactor_,(int)i)) -> Size via cvRound */
    float f = (float)(1.0f / pow((double)sf, (double)i));
    iw[i] = cv_round_f(fw * f);
    ih[i] = cv_round_f(fh * f);
    /* :118 scales_[i] = pow(sfactor_, i) * binsize_ */
    scales[i] = (float)(pow((double)sf, (double)i) * (double)sbin);
    for (int j = i + interval; j < n; j += interval) {
      /* :122 pyrDown -> ((w+1)/2, (h+1)/2) ; :124 scales_[j] = 2*scales_[j-interval] */
      iw[j] = (iw[j - interval] + 1) / 2;
      ih[j] = (ih[j - interval] + 1) / 2;
      scales[j] = 2 * scales[j - interval];
    }
  }
  for (int l = 0; l < n; ++l) orc_cells_of(iw[l], ih[l], sbin, &cw[l], &ch[l]);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* cv::resize(INTER_LINEAR), 8-bit: OpenCV 2.4 imgproc/src/imgwarp.cpp        */
/* (resizeGeneric_ + HResizeLinear + VResizeLinear<uchar,int,short,...>)      */
/* call site src/HOGFeatures.cpp:116                                          */
/* ------------------------------------------------------------------------- */
static short sat_short(float v) {
  int i = cv_round_f(v);
  return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i));
}

ORC_API void orc_resize_linear_8u(const uint8_t* src, int sw, int sh, int cn, int sstride,
                                  uint8_t* dst, int dw, int dh) {
  if (sw == dw && sh == dh) { /* bilinear with all-zero fractions == copy */
    for (int y = 0; y < sh; ++y) memcpy(dst + (size_t)y * dw * cn, src + (size_t)y * sstride, (size_t)sw * cn);
    return;
  }
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int* xofs = (int*)malloc(sizeof(int) * dw);
  short* ialpha = (short*)malloc(sizeof(short) * dw * 2);
  int* yofs = (int*)malloc(sizeof(int) * dh);
  short* ibeta = (short*)malloc(sizeof(short) * dh * 2);
  int xmax = dw;
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) {
      xmax = imin(xmax, dx);
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    }
    xofs[dx] = sx;
    ialpha[dx * 2] = sat_short((1.f - fx) * 2048);
    ialpha[dx * 2 + 1] = sat_short(fx * 2048);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[dy * 2] = sat_short((1.f - fy) * 2048);
    ibeta[dy * 2 + 1] = sat_short(fy * 2048);
  }
  int* rows[2];
  rows[0] = (int*)malloc(sizeof(int) * dw * cn);
  rows[1] = (int*)malloc(sizeof(int) * dw * cn);
  for (int dy = 0; dy < dh; ++dy) {
    for (int k = 0; k < 2; ++k) {
      int sy = yofs[dy] + k;
      sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
      const uint8_t* S = src + (size_t)sy * sstride;
      int* D = rows[k];
      for (int dx = 0; dx < dw; ++dx) {
        int sx = xofs[dx] * cn;
        for (int c = 0; c < cn; ++c) {
          if (dx < xmax)
            D[dx * cn + c] = S[sx + c] * ialpha[dx * 2] + S[sx + cn + c] * ialpha[dx * 2 + 1];
          else
            D[dx * cn + c] = S[sx + c] * 2048;
        }
      }
    }
    short b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t* out = dst + (size_t)dy * dw * cn;
    for (int x = 0; x < dw * cn; ++x) {
      int v = (((b0 * (rows[0][x] >> 4)) >> 16) + ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2;
      out[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  free(rows[0]); free(rows[1]); free(xofs); free(ialpha); free(yofs); free(ibeta);
}

/* ------------------------------------------------------------------------- */
/* cv::pyrDown, 8-bit: OpenCV 2.4 imgproc/src/pyramids.cpp                    */
/* ([1 4 6 4 1] x [1 4 6 4 1], (sum+128)>>8, BORDER_REFLECT_101)              */
/* call site src/HOGFeatures.cpp:122                                          */
/* ------------------------------------------------------------------------- */
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while ((unsigned)p >= (unsigned)len) {
    if (p < 0) p = -p; else p = 2 * len - 2 - p;
  }
  return p;
}

ORC_API void orc_pyrdown_8u(const uint8_t* src, int sw, int sh, int cn, int sstride, uint8_t* dst) {
  const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  static const int wt[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < dh; ++y) {
    for (int x = 0; x < dw; ++x) {
      for (int c = 0; c < cn; ++c) {
        int sum = 0;
        for (int i = 0; i < 5; ++i) {
          int sy = reflect101(2 * y + i - 2, sh);
          int rs = 0;
          for (int j = 0; j < 5; ++j) {
            int sx = reflect101(2 * x + j - 2, sw);
            rs += wt[j] * src[(size_t)sy * sstride + sx * cn + c];
          }
          sum += wt[i] * rs;
        }
        dst[((size_t)y * dw + x) * cn + c] = (uint8_t)((sum + 128) >> 8);
      }
    }
  }
}

/* image pyramid: src/HOGFeatures.cpp:111-127. out = level images back to back */
ORC_API int orc_image_pyramid_u8(const uint8_t* im, int w, int h, int cn, int stride, int sbin,
                                 int interval, uint8_t* out, size_t* offsets) {
  int n;
  int32_t iw[128], ih[128], cw[128], ch[128];
  float sc[128];
  if (orc_pyramid_geometry(w, h, sbin, interval, &n, iw, ih, cw, ch, sc)) return -1;
  size_t off = 0;
  for (int l = 0; l < n; ++l) { offsets[l] = off; off += (size_t)iw[l] * ih[l] * cn; }
  offsets[n] = off;
#ifdef _OPENMP
#pragma omp parallel for
#endif
  for (int i = 0; i < interval; ++i) {
    orc_resize_linear_8u(im, w, h, cn, stride, out + offsets[i], iw[i], ih[i]);
    for (int j = i + interval; j < n; j += interval)
      orc_pyrdown_8u(out + offsets[j - interval], iw[j - interval], ih[j - interval], cn,
                     iw[j - interval] * cn, out + offsets[j]);
  }
  return n;
}

/* ------------------------------------------------------------------------- */
/* HOGFeatures<float>::features<uint8_t>: src/HOGFeatures.cpp:168-341         */
/* ------------------------------------------------------------------------- */
ORC_API int orc_hog_u8(const uint8_t* im, int w, int h, int cn, int stride, int sbin, float* feat) {
  const int norient = 18, flen = 32;
  const int color = (cn == 3);
  if (cn != 1 && cn != 3) return -1;
  const int bw = (int)roundf((float)w / (float)sbin), bh = (int)roundf((float)h / (float)sbin); /* :174 */
  const int ow = imax(bw - 2, 0), oh = imax(bh - 2, 0);                                         /* :175 */
  const int vw = bw * sbin, vh = bh * sbin;                                                     /* :176 */
  float* hist = (float*)calloc((size_t)bw * bh * norient + 1, sizeof(float));
  float* norm = (float*)calloc((size_t)bw * bh + 1, sizeof(float));
  memset(feat, 0, sizeof(float) * (size_t)ow * oh * flen);
  const size_t histstride = (size_t)bw * norient, normstride = bw, featstride = (size_t)ow * flen;
  const double eps = 0.0001;                                                                    /* :189 */
  const float uu[9] = {1.000, 0.9397, 0.7660, 0.5000, 0.1736, -0.1736, -0.5000, -0.7660, -0.9397};
  const float vv[9] = {0.000, 0.3420, 0.6428, 0.8660, 0.9848, 0.9848, 0.8660, 0.6428, 0.3420};

  for (int y = 1; y < vh - 1; ++y) {                                                            /* :202 */
    for (int x = 1; x < vw - 1; ++x) {
      float dx, dy, v;
      const int sx = imin(x, w - 2), sy = imin(y, h - 2);
      if (!color) {                                                                             /* :207 */
        const uint8_t* s = im + sx + (size_t)sy * stride;
        dy = (float)(*(s + stride) - *(s - stride));
        dx = (float)(*(s + 1) - *(s - 1));
        v = dx * dx + dy * dy;
      } else {                                                                                  /* :217 */
        const uint8_t* s = im + 3 * sx + (size_t)sy * stride;
        float dyb = (float)(*(s + stride) - *(s - stride));
        float dxb = (float)(*(s + 3) - *(s - 3));
        float vb = dxb * dxb + dyb * dyb;
        s += 1;
        float dyg = (float)(*(s + stride) - *(s - stride));
        float dxg = (float)(*(s + 3) - *(s - 3));
        float vg = dxg * dxg + dyg * dyg;
        s += 1;
        dy = (float)(*(s + stride) - *(s - stride));
        dx = (float)(*(s + 3) - *(s - 3));
        v = dx * dx + dy * dy;
        if (vg > v) { v = vg; dx = dxg; dy = dyg; }                                             /* :238 */
        if (vb > v) { v = vb; dx = dxb; dy = dyb; }
      }
      float best_dot = 0;                                                                       /* :243 */
      int best_o = 0;
      for (int o = 0; o < norient / 2; ++o) {
        float dot = uu[o] * dx + vv[o] * dy;
        if (dot > best_dot) { best_dot = dot; best_o = o; }
        else if (-dot > best_dot) { best_dot = -dot; best_o = o + norient / 2; }
      }
      /* :252-260 — literals are double: evaluated in double, narrowed to T */
      float yp = (float)(((double)(float)y + 0.5) / (double)(float)sbin - 0.5);
      float xp = (float)(((double)(float)x + 0.5) / (double)(float)sbin - 0.5);
      int iyp = (int)floorf(yp);
      int ixp = (int)floorf(xp);
      float vy0 = yp - (float)iyp;
      float vx0 = xp - (float)ixp;
      float vy1 = (float)(1.0 - (double)vy0);
      float vx1 = (float)(1.0 - (double)vx0);
      v = sqrtf(v);
      if (iyp >= 0 && ixp >= 0) *(hist + iyp * histstride + ixp * norient + best_o) += vy1 * vx1 * v;
      if (iyp >= 0 && ixp + 1 < bw) *(hist + iyp * histstride + (ixp + 1) * norient + best_o) += vx0 * vy1 * v;
      if (iyp + 1 < bh && ixp >= 0) *(hist + (iyp + 1) * histstride + ixp * norient + best_o) += vy0 * vx1 * v;
      if (iyp + 1 < bh && ixp + 1 < bw) *(hist + (iyp + 1) * histstride + (ixp + 1) * norient + best_o) += vy0 * vx0 * v;
    }
  }
  for (int y = 0; y < bh; ++y) {                                                                /* :270 */
    const float* src = hist + y * histstride;
    float* dst = norm + y * normstride;
    for (int x = 0; x < bw; ++x) {
      float acc = 0;
      for (int o = 0; o < norient / 2; ++o) {
        float t = *src + *(src + norient / 2);
        acc += t * t;
        src++;
      }
      *dst++ = acc;
      src += norient / 2;
    }
  }
  for (int y = 0; y < oh; ++y) {                                                                /* :286 */
    for (int x = 0; x < ow; ++x) {
      float* dst = feat + y * featstride + (size_t)x * flen;
      const float* p;
      float n1, n2, n3, n4;
      p = norm + (y + 1) * normstride + (x + 1);
      n1 = (float)(1.0f / sqrt((double)(*p + *(p + 1) + *(p + normstride) + *(p + normstride + 1)) + eps));
      p = norm + y * normstride + (x + 1);
      n2 = (float)(1.0f / sqrt((double)(*p + *(p + 1) + *(p + normstride) + *(p + normstride + 1)) + eps));
      p = norm + (y + 1) * normstride + x;
      n3 = (float)(1.0f / sqrt((double)(*p + *(p + 1) + *(p + normstride) + *(p + normstride + 1)) + eps));
      p = norm + y * normstride + x;
      n4 = (float)(1.0f / sqrt((double)(*p + *(p + 1) + *(p + normstride) + *(p + normstride + 1)) + eps));
      float t1 = 0, t2 = 0, t3 = 0, t4 = 0;
      const float* src = hist + (y + 1) * histstride + (size_t)(x + 1) * norient;
      for (int o = 0; o < norient; ++o) {                                                       /* :305 */
        float val = *src;
        float h1 = fminf(val * n1, 0.2f), h2 = fminf(val * n2, 0.2f);
        float h3 = fminf(val * n3, 0.2f), h4 = fminf(val * n4, 0.2f);
        *(dst++) = (float)(0.5 * (double)(h1 + h2 + h3 + h4));
        src++;
        t1 += h1; t2 += h2; t3 += h3; t4 += h4;
      }
      src = hist + (y + 1) * histstride + (size_t)(x + 1) * norient;
      for (int o = 0; o < norient / 2; ++o) {                                                   /* :320 */
        float sum = *src + *(src + norient / 2);
        float h1 = fminf(sum * n1, 0.2f), h2 = fminf(sum * n2, 0.2f);
        float h3 = fminf(sum * n3, 0.2f), h4 = fminf(sum * n4, 0.2f);
        *(dst++) = (float)(0.5 * (double)(h1 + h2 + h3 + h4));
        src++;
      }
      *(dst++) = (float)(0.2357 * (double)t1);                                                  /* :331 */
      *(dst++) = (float)(0.2357 * (double)t2);
      *(dst++) = (float)(0.2357 * (double)t3);
      *(dst++) = (float)(0.2357 * (double)t4);
      *dst = 0;                                                                                 /* :337 */
    }
  }
  free(hist); free(norm);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* SpatialConvolutionEngine::convolve + Filter2D:                             */
/* src/SpatialConvolutionEngine.cpp:70-94,133-159; src/filter.cpp:3808-3924    */
/* "same" correlation, anchor = kernel centre, constant border 0 (1 for the    */
/* last channel), per-channel tap-ordered mul+add in float, then pdf += pdfc.  */
/* ------------------------------------------------------------------------- */
ORC_API void orc_pdf_one(const float* feat, int H, int W, int flen, const float* filt, int kh, int kw,
                         float* out) {
  const int ay = kh / 2, ax = kw / 2; /* normalizeAnchor(Point(-1,-1)) */
  float* pdfc = (float*)malloc(sizeof(float) * (size_t)H * W);
  for (size_t i = 0; i < (size_t)H * W; ++i) out[i] = 0.0f; /* :81 Mat::zeros */
  for (int c = 0; c < flen; ++c) {
    const float border = (c == flen - 1) ? 1.0f : 0.0f; /* :147-155 */
    for (int y = 0; y < H; ++y) {
      for (int x = 0; x < W; ++x) {
        float s = 0.0f; /* delta */
        for (int i = 0; i < kh; ++i) {
          for (int j = 0; j < kw; ++j) {
            float f = filt[((size_t)i * kw + j) * flen + c];
            if (f == 0) continue; /* preprocess2DKernel keeps non-zero taps only */
            int yy = y + i - ay, xx = x + j - ax;
            float v = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? border
                                                               : feat[((size_t)yy * W + xx) * flen + c];
            s += f * v;
          }
        }
        pdfc[(size_t)y * W + x] = s;
      }
    }
    for (size_t i = 0; i < (size_t)H * W; ++i) out[i] += pdfc[i]; /* :92 */
  }
  free(pdfc);
}

/* SpatialConvolutionEngine::pdf for one level: out[nf][H][W] (:106-124) */
ORC_API void orc_pdf_level(const float* feat, int H, int W, int flen, const float* filters, int nf,
                           int kh, int kw, float* out) {
#ifdef _OPENMP
#pragma omp parallel for
#endif
  for (int n = 0; n < nf; ++n)
    orc_pdf_one(feat, H, W, flen, filters + (size_t)n * kh * kw * flen, kh, kw, out + (size_t)n * H * W);
}

/* ------------------------------------------------------------------------- */
/* DistanceTransform<float>: include/DistanceTransform.hpp:89-105,151-245      */
/* ------------------------------------------------------------------------- */
static inline double quad_isect(double a, double b, int x0, int x1, double y0, double y1) {
  /* :98-100 */
  return ((y1 - y0) - b * (x1 - x0) + a * (x1 * x1 - x0 * x0)) / (2 * a * (x1 - x0));
}
static inline double quad_eval(double a, double b, int x, double y) { return a * (x * x) + b * x + y; } /* :103 */

ORC_API void orc_dt1d(const float* src, float* dst, int32_t* ptr, int N, double a, double b, int os) {
  int* v = (int*)malloc(sizeof(int) * (N > 0 ? N : 1));
  float* z = (float*)malloc(sizeof(float) * (N + 1));
  int k = 0;
  v[0] = 0;
  z[0] = -INFINITY;
  z[1] = +INFINITY;
  for (int q = 1; q < N; ++q) {
    float s = (float)quad_isect(a, b, v[k], q, src[v[k]], src[q]);
    while (s <= z[k] && k > 0) {
      k--;
      s = (float)quad_isect(a, b, v[k], q, src[v[k]], src[q]);
    }
    k++;
    v[k] = q;
    z[k] = s;
    z[k + 1] = +INFINITY;
  }
  k = 0;
  for (int q = 0; q < N; ++q) {
    while (z[k + 1] < (float)os) k++;
    dst[q] = (float)quad_eval(a, b, os - v[k], src[v[k]]);
    ptr[q] = v[k];
    os++;
  }
  free(v); free(z);
}

/* compute(): x pass over rows, y pass over columns, then the pointer
 * composition Iy'(m,n) = Iy(m, Ix(m,n)) of :233-244 (correct_ptr=0), or the
 * true arg-max composition Ix'(m,n)=Ix(Iy(m,n),n) (correct_ptr=1, not in the
 * reference).                                                 */
ORC_API void orc_dt2d(const float* in, int M, int N, double ax, double bx, double ay, double by,
                      int osx, int osy, float* out, int32_t* Ix, int32_t* Iy, int correct_ptr) {
  float* tmp = (float*)malloc(sizeof(float) * (size_t)M * N);
  float* col = (float*)malloc(sizeof(float) * M);
  float* cold = (float*)malloc(sizeof(float) * M);
  int32_t* colp = (int32_t*)malloc(sizeof(int32_t) * M);
  for (int m = 0; m < M; ++m) orc_dt1d(in + (size_t)m * N, tmp + (size_t)m * N, Ix + (size_t)m * N, N, ax, bx, osx);
  for (int n = 0; n < N; ++n) {
    for (int m = 0; m < M; ++m) col[m] = tmp[(size_t)m * N + n];
    orc_dt1d(col, cold, colp, M, ay, by, osy);
    for (int m = 0; m < M; ++m) { out[(size_t)m * N + n] = cold[m]; Iy[(size_t)m * N + n] = colp[m]; }
  }
  int32_t* row = (int32_t*)malloc(sizeof(int32_t) * N);
  if (!correct_ptr) {
    for (int m = 0; m < M; ++m) {
      for (int n = 0; n < N; ++n) row[n] = Iy[(size_t)m * N + Ix[(size_t)m * N + n]];
      for (int n = 0; n < N; ++n) Iy[(size_t)m * N + n] = row[n];
    }
  } else {
    int32_t* nix = (int32_t*)malloc(sizeof(int32_t) * (size_t)M * N);
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) nix[(size_t)m * N + n] = Ix[(size_t)Iy[(size_t)m * N + n] * N + n];
    memcpy(Ix, nix, sizeof(int32_t) * (size_t)M * N);
    free(nix);
  }
  free(row); free(tmp); free(col); free(cold); free(colp);
}

/* ------------------------------------------------------------------------- */
/* DynamicProgram<float>::min for one (level, component):                     */
/* src/DynamicProgram.cpp:66-173, include/Math.hpp:108-185, include/Parts.hpp  */
/* resp: [nfilters][H][W].  Pointer outputs: for flat part fp of this          */
/* component (p>=1) and parent mixture m the plane index is                    */
/*   ptr_plane(c,p,m) = sum_{p'<p, p'>=1} L(p') + m,  L = #mixtures of parent. */
/* ------------------------------------------------------------------------- */
static int nmix_of(const pbd_model_desc* md, int fp) { return md->mix_offset[fp + 1] - md->mix_offset[fp]; }

ORC_API int orc_ptr_planes(const pbd_model_desc* md, int comp) {
  int p0 = md->part_offset[comp], p1 = md->part_offset[comp + 1], n = 0;
  for (int fp = p0 + 1; fp < p1; ++fp) n += nmix_of(md, p0 + md->parentid[fp]);
  return n;
}

ORC_API int orc_dp_min_level(const pbd_model_desc* md, int comp, const float* resp, int H, int W,
                             int32_t* Ix, int32_t* Iy, int32_t* Ik, float* rootv, int32_t* rooti,
                             int correct_ptr) {
  const size_t HW = (size_t)H * W;
  const int p0 = md->part_offset[comp], np = md->part_offset[comp + 1] - p0;
  float** nc = (float**)calloc(md->nfilters, sizeof(float*)); /* ncscores: NULL == empty() */
  /* plane offsets */
  int* plane0 = (int*)malloc(sizeof(int) * np);
  { int acc = 0; for (int p = 1; p < np; ++p) { plane0[p] = acc; acc += nmix_of(md, p0 + md->parentid[p0 + p]); } }
  for (int p = np - 1; p > 0; --p) { /* :95 */
    const int fp = p0 + p, fpar = p0 + md->parentid[fp];
    const int K = nmix_of(md, fp), L = nmix_of(md, fpar);
    const int fm0 = md->mix_offset[fp], pm0 = md->mix_offset[fpar];
    float* sdt = (float*)malloc(sizeof(float) * HW * K);
    int32_t* ixp = (int32_t*)malloc(sizeof(int32_t) * HW * K);
    int32_t* iyp = (int32_t*)malloc(sizeof(int32_t) * HW * K);
    for (int m = 0; m < K; ++m) { /* :110-132 */
      const int fid = md->filterid[fm0 + m];
      const float* score_in = nc[fid] ? nc[fid] : resp + (size_t)fid * HW;
      const int did = md->defid[fm0 + m];
      const float* w = md->defw + (size_t)did * 4;
      orc_dt2d(score_in, H, W, -(double)w[0], -(double)w[1], -(double)w[2], -(double)w[3],
               md->anchors[did * 2], md->anchors[did * 2 + 1], sdt + HW * m, ixp + HW * m, iyp + HW * m,
               correct_ptr);
    }
    for (int m = 0; m < L; ++m) { /* :134-160 */
      int32_t* oix = Ix + (size_t)(plane0[p] + m) * HW;
      int32_t* oiy = Iy + (size_t)(plane0[p] + m) * HW;
      int32_t* oik = Ik + (size_t)(plane0[p] + m) * HW;
      const int pfid = md->filterid[pm0 + m];
      if (!nc[pfid]) { /* :155 lazily copy the parent's raw response */
        nc[pfid] = (float*)malloc(sizeof(float) * HW);
        memcpy(nc[pfid], resp + (size_t)pfid * HW, sizeof(float) * HW);
      }
      for (size_t i = 0; i < HW; ++i) {
        float v = -INFINITY; /* Math::reduceMax :176-182, strict > : first max wins */
        int bi = 0;
        if (K == 1) { /* :154-158 K==1 shortcut: copy, index 0 */
          v = sdt[i] + md->biasw[md->biasid[fm0] + m];
        } else {
          for (int mm = 0; mm < K; ++mm) {
            float wv = sdt[HW * mm + i] + md->biasw[md->biasid[fm0 + mm] + m]; /* :139 */
            if (wv > v) { bi = mm; v = wv; }
          }
        }
        oik[i] = bi;
        oix[i] = ixp[HW * bi + i]; /* reducePickIndex :130 */
        oiy[i] = iyp[HW * bi + i];
        nc[pfid][i] += v; /* :156 */
      }
    }
    free(sdt); free(ixp); free(iyp);
  }
  { /* root :163-171 */
    const int K0 = nmix_of(md, p0), fm0 = md->mix_offset[p0];
    const float bias = md->biasw[md->biasid[fm0]]; /* root.bias(0)[0] */
    for (size_t i = 0; i < HW; ++i) {
      float v = -INFINITY;
      int bi = 0;
      for (int m = 0; m < K0; ++m) {
        const int fid = md->filterid[fm0 + m];
        /* NB: root.score(ncscores,m) is an EMPTY Mat if the root mixture never received a
         * message; every model here gives the root children for all mixtures.              */
        const float* sc = nc[fid] ? nc[fid] : resp + (size_t)fid * HW;
        float wv = sc[i] + bias;
        if (K0 == 1) { v = wv; bi = 0; break; }
        if (wv > v) { bi = m; v = wv; }
      }
      rootv[i] = v;
      rooti[i] = bi;
    }
  }
  for (int f = 0; f < md->nfilters; ++f) free(nc[f]);
  free(nc); free(plane0);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* DynamicProgram<float>::argmin for one (level, component):                  */
/* src/DynamicProgram.cpp:189-255.  Appends to heads/boxes/locs at *count.     */
/* ------------------------------------------------------------------------- */
ORC_API int orc_dp_argmin_level(const pbd_model_desc* md, int comp, int level, float scale,
                                const float* rootv, const int32_t* rooti, const int32_t* Ix,
                                const int32_t* Iy, const int32_t* Ik, int H, int W, int max_parts,
                                pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity,
                                int* count) {
  const size_t HW = (size_t)H * W;
  const int p0 = md->part_offset[comp], np = md->part_offset[comp + 1] - p0;
  int* plane0 = (int*)malloc(sizeof(int) * np);
  { int acc = 0; for (int p = 1; p < np; ++p) { plane0[p] = acc; acc += nmix_of(md, p0 + md->parentid[p0 + p]); } }
  int* xv = (int*)malloc(sizeof(int) * np), *yv = (int*)malloc(sizeof(int) * np), *mv = (int*)malloc(sizeof(int) * np);
  const double thresh = (double)md->thresh; /* DynamicProgram(double thresh) from float Model::thresh() */
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) { /* Math::find: row-major */
      if (!((double)rootv[(size_t)y * W + x] > thresh)) continue; /* :208 strict > */
      int idx = *count;
      (*count)++;
      if (idx >= capacity) continue;
      heads[idx].score = rootv[(size_t)y * W + x];
      heads[idx].component = comp;
      heads[idx].level = level;
      heads[idx].nparts = np;
      for (int p = 0; p < np; ++p) {
        if (p == 0) { xv[0] = x; yv[0] = y; mv[0] = rooti[(size_t)y * W + x]; }
        else {
          int par = md->parentid[p0 + p];
          int px = xv[par], py = yv[par], pm = mv[par];
          size_t o = (size_t)(plane0[p] + pm) * HW + (size_t)py * W + px;
          xv[p] = Ix[o]; yv[p] = Iy[o]; mv[p] = Ik[o];
        }
        /* :238-240: Point*float rounds each coordinate with cvRound; xsize=ysize=filter.rows */
        int x1 = cv_round_f((float)(xv[p] - 1) * scale), y1 = cv_round_f((float)(yv[p] - 1) * scale);
        int sz = cv_round_f((float)md->kh * scale);
        int x2 = x1 + sz - 1, y2 = y1 + sz - 1;
        int32_t* b = boxes ? boxes + ((size_t)idx * max_parts + p) * 4 : 0;
        if (b) { b[0] = imin(x1, x2); b[1] = imin(y1, y2); b[2] = imax(x1, x2) - b[0]; b[3] = imax(y1, y2) - b[1]; }
        int32_t* lc = locs ? locs + ((size_t)idx * max_parts + p) * 3 : 0;
        if (lc) { lc[0] = xv[p]; lc[1] = yv[p]; lc[2] = mv[p]; }
      }
    }
  }
  free(plane0); free(xv); free(yv); free(mv);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* PartsBasedDetector<float>::detect: src/PartsBasedDetector.cpp:69-95         */
/* OpenMP at the reference's five sites.  stage_ms[5] =          */
/* {image pyramid, hog, pdf, dp min, argmin}.  keep = optional intermediates.  */
/* ------------------------------------------------------------------------- */
typedef struct orc_frame {
  int nlevels, cn;
  int32_t iw[128], ih[128], cw[128], ch[128];
  float scales[128];
  uint8_t* pyr; size_t pyr_off[129];
  float* feat[128];
  float* resp[128];           /* [nf][H][W] */
  int32_t* Ix[128]; int32_t* Iy[128]; int32_t* Ik[128]; /* per level, components back to back */
  float* rootv[128]; int32_t* rooti[128];               /* [ncomp][H][W] */
} orc_frame;

ORC_API void orc_frame_free(orc_frame* f) {
  if (!f) return;
  free(f->pyr);
  for (int l = 0; l < f->nlevels; ++l) {
    free(f->feat[l]); free(f->resp[l]); free(f->Ix[l]); free(f->Iy[l]); free(f->Ik[l]);
    free(f->rootv[l]); free(f->rooti[l]);
  }
  free(f);
}

ORC_API int orc_max_parts(const pbd_model_desc* md) {
  int mx = 0;
  for (int c = 0; c < md->ncomponents; ++c) mx = imax(mx, md->part_offset[c + 1] - md->part_offset[c]);
  return mx;
}

ORC_API int orc_detect_u8(const pbd_model_desc* md, const uint8_t* im, int w, int h, int cn, int stride,
                          pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity,
                          int* count, double* stage_ms, orc_frame** keep, int correct_ptr) {
  orc_frame* f = (orc_frame*)calloc(1, sizeof(orc_frame));
  double t0 = now_ms();
  if (orc_pyramid_geometry(w, h, md->sbin, md->interval, &f->nlevels, f->iw, f->ih, f->cw, f->ch, f->scales)) {
    free(f);
    return -1;
  }
  const int n = f->nlevels;
  f->cn = cn;
  size_t tot = 0;
  for (int l = 0; l < n; ++l) tot += (size_t)f->iw[l] * f->ih[l] * cn;
  f->pyr = (uint8_t*)malloc(tot);
  orc_image_pyramid_u8(im, w, h, cn, stride, md->sbin, md->interval, f->pyr, f->pyr_off);
  double t1 = now_ms();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (int l = 0; l < n; ++l) { /* src/HOGFeatures.cpp:130-150 */
    f->feat[l] = (float*)malloc(sizeof(float) * ((size_t)f->cw[l] * f->ch[l] * md->flen + 1));
    orc_hog_u8(f->pyr + f->pyr_off[l], f->iw[l], f->ih[l], cn, f->iw[l] * cn, md->sbin, f->feat[l]);
  }
  double t2 = now_ms();
  for (int l = 0; l < n; ++l) f->resp[l] = (float*)malloc(sizeof(float) * ((size_t)md->nfilters * f->cw[l] * f->ch[l] + 1));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (int nf = 0; nf < md->nfilters; ++nf) /* src/SpatialConvolutionEngine.cpp:114-123 */
    for (int l = 0; l < n; ++l)
      orc_pdf_one(f->feat[l], f->ch[l], f->cw[l], md->flen,
                  md->filters + (size_t)nf * md->kh * md->kw * md->flen, md->kh, md->kw,
                  f->resp[l] + (size_t)nf * f->cw[l] * f->ch[l]);
  double t3 = now_ms();
  int planes_tot = 0;
  int* comp_plane0 = (int*)malloc(sizeof(int) * (md->ncomponents + 1));
  for (int c = 0; c < md->ncomponents; ++c) { comp_plane0[c] = planes_tot; planes_tot += orc_ptr_planes(md, c); }
  for (int l = 0; l < n; ++l) {
    size_t HW = (size_t)f->cw[l] * f->ch[l];
    f->Ix[l] = (int32_t*)malloc(sizeof(int32_t) * (HW * planes_tot + 1));
    f->Iy[l] = (int32_t*)malloc(sizeof(int32_t) * (HW * planes_tot + 1));
    f->Ik[l] = (int32_t*)malloc(sizeof(int32_t) * (HW * planes_tot + 1));
    f->rootv[l] = (float*)malloc(sizeof(float) * (HW * md->ncomponents + 1));
    f->rooti[l] = (int32_t*)malloc(sizeof(int32_t) * (HW * md->ncomponents + 1));
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (int nc = 0; nc < n * md->ncomponents; ++nc) { /* src/DynamicProgram.cpp:80-87 */
    int l = nc / md->ncomponents, c = nc % md->ncomponents;
    size_t HW = (size_t)f->cw[l] * f->ch[l];
    orc_dp_min_level(md, c, f->resp[l], f->ch[l], f->cw[l], f->Ix[l] + HW * comp_plane0[c],
                     f->Iy[l] + HW * comp_plane0[c], f->Ik[l] + HW * comp_plane0[c],
                     f->rootv[l] + HW * c, f->rooti[l] + HW * c, correct_ptr);
  }
  double t4 = now_ms();
  *count = 0;
  const int mp = orc_max_parts(md);
  for (int l = 0; l < n; ++l) /* single-threaded order of src/DynamicProgram.cpp:197-253 */
    for (int c = 0; c < md->ncomponents; ++c) {
      size_t HW = (size_t)f->cw[l] * f->ch[l];
      orc_dp_argmin_level(md, c, l, f->scales[l], f->rootv[l] + HW * c, f->rooti[l] + HW * c,
                          f->Ix[l] + HW * comp_plane0[c], f->Iy[l] + HW * comp_plane0[c],
                          f->Ik[l] + HW * comp_plane0[c], f->ch[l], f->cw[l], mp, heads, boxes, locs,
                          capacity, count);
    }
  double t5 = now_ms();
  free(comp_plane0);
  if (stage_ms) { stage_ms[0] = t1 - t0; stage_ms[1] = t2 - t1; stage_ms[2] = t3 - t2; stage_ms[3] = t4 - t3; stage_ms[4] = t5 - t4; }
  if (keep) *keep = f; else orc_frame_free(f);
  return 0;
}

/* accessors for kept intermediates (ctypes-friendly) */
ORC_API int orc_frame_nlevels(const orc_frame* f) { return f->nlevels; }
ORC_API const uint8_t* orc_frame_image(const orc_frame* f, int l) { return f->pyr + f->pyr_off[l]; }
ORC_API const float* orc_frame_feat(const orc_frame* f, int l) { return f->feat[l]; }
ORC_API const float* orc_frame_resp(const orc_frame* f, int l) { return f->resp[l]; }
ORC_API const int32_t* orc_frame_ix(const orc_frame* f, int l) { return f->Ix[l]; }
ORC_API const int32_t* orc_frame_iy(const orc_frame* f, int l) { return f->Iy[l]; }
ORC_API const int32_t* orc_frame_ik(const orc_frame* f, int l) { return f->Ik[l]; }
ORC_API const float* orc_frame_rootv(const orc_frame* f, int l) { return f->rootv[l]; }
ORC_API const int32_t* orc_frame_rooti(const orc_frame* f, int l) { return f->rooti[l]; }
ORC_API void orc_frame_dims(const orc_frame* f, int l, int* iw, int* ih, int* cw, int* ch, float* scale) {
  *iw = f->iw[l]; *ih = f->ih[l]; *cw = f->cw[l]; *ch = f->ch[l]; *scale = f->scales[l];
}

/* ------------------------------------------------------------------------- */
/* Candidate::sort / Candidate::nonMaximaSuppression:                         */
/* include/Candidate.hpp:91-111,277-304 (sort made stable: std::sort leaves    */
/* the order of equal scores unspecified)                                      */
/* ------------------------------------------------------------------------- */
ORC_API int orc_candidates_sort(pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int count, int mp) {
  int* order = (int*)malloc(sizeof(int) * (count + 1));
  for (int i = 0; i < count; ++i) order[i] = i;
  for (int i = 1; i < count; ++i) { /* stable insertion sort, descending score */
    int o = order[i], j = i - 1;
    while (j >= 0 && heads[order[j]].score < heads[o].score) { order[j + 1] = order[j]; --j; }
    order[j + 1] = o;
  }
  pbd_candidate_head* h2 = (pbd_candidate_head*)malloc(sizeof(*h2) * (count + 1));
  int32_t* b2 = boxes ? (int32_t*)malloc(sizeof(int32_t) * ((size_t)count * mp * 4 + 1)) : 0;
  int32_t* l2 = locs ? (int32_t*)malloc(sizeof(int32_t) * ((size_t)count * mp * 3 + 1)) : 0;
  for (int i = 0; i < count; ++i) {
    h2[i] = heads[order[i]];
    if (b2) memcpy(b2 + (size_t)i * mp * 4, boxes + (size_t)order[i] * mp * 4, sizeof(int32_t) * mp * 4);
    if (l2) memcpy(l2 + (size_t)i * mp * 3, locs + (size_t)order[i] * mp * 3, sizeof(int32_t) * mp * 3);
  }
  memcpy(heads, h2, sizeof(*h2) * count);
  if (b2) { memcpy(boxes, b2, sizeof(int32_t) * (size_t)count * mp * 4); free(b2); }
  if (l2) { memcpy(locs, l2, sizeof(int32_t) * (size_t)count * mp * 3); free(l2); }
  free(h2); free(order);
  return 0;
}

ORC_API int orc_candidates_nms(pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int count,
                               int mp, int im_w, int im_h, float overlap, int* kept) {
  uint8_t* scratch = (uint8_t*)calloc((size_t)im_w * im_h + 1, 1);
  int keep = 0;
  for (int n = 0; n < count; ++n) {
    /* boundingBox(): union of the part rects (cv::Rect |, OpenCV 2.4 semantics) */
    const int32_t* b = boxes + (size_t)n * mp * 4;
    int x = b[0], y = b[1], bw = b[2], bh = b[3];
    for (int p = 0; p < heads[n].nparts; ++p) {
      const int32_t* q = b + p * 4;
      int x1 = imin(x, q[0]), y1 = imin(y, q[1]);
      bw = imax(x + bw, q[0] + q[2]) - x1;
      bh = imax(y + bh, q[1] + q[3]) - y1;
      x = x1; y = y1;
    }
    /* & bounds */
    int ix1 = imax(x, 0), iy1 = imax(y, 0);
    int iw = imin(x + bw, im_w) - ix1, ih = imin(y + bh, im_h) - iy1;
    if (iw <= 0 || ih <= 0) { ix1 = iy1 = iw = ih = 0; }
    double sum = 0;
    for (int yy = iy1; yy < iy1 + ih; ++yy)
      for (int xx = ix1; xx < ix1 + iw; ++xx) sum += scratch[(size_t)yy * im_w + xx];
    if (sum / (double)(iw * ih) > (double)overlap) continue; /* :296 (0/0 = NaN keeps) */
    for (int yy = iy1; yy < iy1 + ih; ++yy) memset(scratch + (size_t)yy * im_w + ix1, 1, iw);
    if (keep != n) {
      heads[keep] = heads[n];
      memmove(boxes + (size_t)keep * mp * 4, boxes + (size_t)n * mp * 4, sizeof(int32_t) * mp * 4);
      if (locs) memmove(locs + (size_t)keep * mp * 3, locs + (size_t)n * mp * 3, sizeof(int32_t) * mp * 3);
    }
    keep++;
  }
  free(scratch);
  *kept = keep;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* nonMaximaSuppression(src, sz, dst, mask=empty): src/nms.cpp:84-129          */
/* (dead code in the reference; restated for the optional GPU pre-filter)      */
/* ------------------------------------------------------------------------- */
ORC_API void orc_nms_map(const float* src, int M, int N, int sz, uint8_t* dst) {
  memset(dst, 0, (size_t)M * N);
  for (int m = 0; m < M; m += sz + 1) {
    for (int n = 0; n < N; n += sz + 1) {
      int i1 = imin(m + sz + 1, M), j1 = imin(n + sz + 1, N);
      /* minMaxLoc: first maximum in row-major order */
      double vcmax = -DBL_MAX; int ci = m, cj = n;
      for (int i = m; i < i1; ++i)
        for (int j = n; j < j1; ++j)
          if ((double)src[(size_t)i * N + j] > vcmax) { vcmax = src[(size_t)i * N + j]; ci = i; cj = j; }
      int in0 = imax(ci - sz, 0), in1 = imin(ci + sz + 1, M);
      int jn0 = imax(cj - sz, 0), jn1 = imin(cj + sz + 1, N);
      /* neighbourhood minus the (sz+1)^2 window starting at the block origin (:113-117) */
      int is0 = m - in0, is1 = imin(m - in0 + sz + 1, in1 - in0);
      int js0 = n - jn0, js1 = imin(n - jn0 + sz + 1, jn1 - jn0);
      double vnmax = -DBL_MAX; int any = 0;
      for (int i = in0; i < in1; ++i)
        for (int j = jn0; j < jn1; ++j) {
          int li = i - in0, lj = j - jn0;
          if (li >= is0 && li < is1 && lj >= js0 && lj < js1) continue;
          any = 1;
          if ((double)src[(size_t)i * N + j] > vnmax) vnmax = src[(size_t)i * N + j];
        }
      /* minMaxLoc over an all-zero mask leaves maxVal = 0 in OpenCV 2.4 */
      if (!any) vnmax = 0;
      if (vcmax > vnmax) dst[(size_t)ci * N + cj] = 255;
    }
  }
}

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
