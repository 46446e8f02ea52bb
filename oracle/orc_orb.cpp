// orc_orb.cpp -- CPU restatement of the ORB front end behind compute_bow_vector
// (include/visnav/keypoints.h:243-254: cv::ORB::create(num_features, 1.2, 8, 19, 0, 2, cv::ORB::FAST_SCORE)
// ->detectAndCompute).  TEST INFRASTRUCTURE ONLY (see vslam_oracle.h).
//
// PARITY UNPINNED: cv::ORB is [upstream] OpenCV 4.x (features2d/src/orb.cpp, fast.cpp, imgproc resize /
// GaussianBlur); the submodule is empty and the reference holds no fixture for it.  What follows restates
// the published algorithm with every arithmetic convention spelled out, so that the HIP kernels can be
// held bit-exact to THIS definition:
//   * pyramid: level l has size (cvRound(w / s_l), cvRound(h / s_l)), s_l = (float)pow((double)1.2f, l); each
//     level is resized from the previous one with OpenCV's 8-bit INTER_LINEAR fixed-point arithmetic
//     (11-bit weights; sample position (d + 0.5) * (src / dst) - 0.5);
//   * FAST-9/16, threshold 20, score = the largest threshold at which the pixel is still a corner, 3x3
//     non-maximum suppression with strict comparisons, scanned over [3, w-3) x [3, h-3);
//   * border filter (edgeThreshold 19) then retainBest(n_level): everything at or above the n-th best score;
//     per-level quotas n_level as in orb.cpp (geometric series, cvRound, remainder to the last level);
//     output order: level, then row, then column;
//   * orientation: intensity centroid over the radius-15 disc (umax table), angle = fastAtan2(m01, m10) in
//     degrees (OpenCV's 7th-order polynomial, fp32);
//   * descriptors: 7x7 Gaussian blur (sigma 2, fp32 separable, BORDER_REFLECT_101, round half to even) of
//     every level, then 256 rotated tests of bit_pattern_31_ (= the reference's own pattern table,
//     keypoints.h:55-131): (x, y) -> (cvRound(x a - y b), cvRound(x b + y a)), a = (float)cos, b = (float)sin of
//     the angle in radians (fp32 angle * (float)(CV_PI / 180)); bit k of byte j = test 8 j + k.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "vslam_oracle.h"

namespace {

struct PatRow {
  int xa, ya, xb, yb;
};
const PatRow kPattern[256] = {
#include "rbrief_pattern.inc"
};

inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_roundf(float v) { return (int)std::lrintf(v); }
inline int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// OpenCV resize, CV_8U, INTER_LINEAR: INTER_RESIZE_COEF_BITS = 11
void resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)std::floor(fx);
    fx -= sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx >= sw - 1) {
      fx = 0;
      sx = sw - 1;
    }
    xofs[dx] = sx;
    ialpha[2 * dx] = (short)cv_roundf((1.f - fx) * 2048.f);
    ialpha[2 * dx + 1] = (short)cv_roundf(fx * 2048.f);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)std::floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = (short)cv_roundf((1.f - fy) * 2048.f);
    ibeta[2 * dy + 1] = (short)cv_roundf(fy * 2048.f);
  }
  for (int dy = 0; dy < dh; dy++) {
    const int sy0 = std::min(std::max(yofs[dy], 0), sh - 1), sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
    const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    for (int dx = 0; dx < dw; dx++) {
      const int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
      const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
      const int S0 = src[(size_t)sy0 * sw + sx] * a0 + src[(size_t)sy0 * sw + sx1] * a1;
      const int S1 = src[(size_t)sy1 * sw + sx] * a0 + src[(size_t)sy1 * sw + sx1] * a1;
      const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
      dst[(size_t)dy * dw + dx] = (uint8_t)std::min(std::max(v, 0), 255);
    }
  }
}

const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// largest t >= thr such that 9 contiguous circle pixels are all > v + t or all < v - t; 0 if not a corner at thr
int fast_score(const uint8_t* img, int w, int x, int y, int thr) {
  const int v = img[(size_t)y * w + x];
  int d[16];
  for (int k = 0; k < 16; k++) d[k] = (int)img[(size_t)(y + kCircle[k][1]) * w + x + kCircle[k][0]] - v;
  int best = -1;  // max over arcs of min over the arc of (+d) resp. (-d)
  for (int s = 0; s < 16; s++) {
    int mn = 1 << 20, mx = -(1 << 20);
    for (int k = 0; k < 9; k++) {
      const int dv = d[(s + k) & 15];
      mn = std::min(mn, dv);
      mx = std::max(mx, dv);
    }
    best = std::max(best, std::max(mn, -mx));
  }
  // corner at threshold t  <=>  best > t;  score = best - 1
  return best > thr ? best - 1 : 0;
}

void gauss7(const uint8_t* src, int w, int h, uint8_t* dst) {
  float k[7];
  {
    double kd[7], sum = 0;
    for (int i = 0; i < 7; i++) {
      const double x = i - 3;
      kd[i] = std::exp(-x * x / (2.0 * 2.0 * 2.0));
      sum += kd[i];
    }
    for (int i = 0; i < 7; i++) k[i] = (float)(kd[i] / sum);
  }
  std::vector<float> tmp((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float s = 0.f;
      for (int i = 0; i < 7; i++) s = s + k[i] * (float)src[(size_t)y * w + reflect101(x + i - 3, w)];
      tmp[(size_t)y * w + x] = s;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float s = 0.f;
      for (int i = 0; i < 7; i++) s = s + k[i] * tmp[(size_t)reflect101(y + i - 3, h) * w + x];
      const int v = cv_roundf(s);
      dst[(size_t)y * w + x] = (uint8_t)std::min(std::max(v, 0), 255);
    }
}

// OpenCV fastAtan2 (degrees)
float fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI),
              p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.220446049250313e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.220446049250313e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

}  // namespace

extern "C" {

void orc_orb_level_sizes(int w, int h, int nlevels, int* lw, int* lh, float* scale) {
  for (int l = 0; l < nlevels; l++) {
    const float s = (float)std::pow((double)1.2f, (double)l);
    scale[l] = s;
    lw[l] = cv_roundf((float)w / s);
    lh[l] = cv_roundf((float)h / s);
  }
}

void orc_orb_level_quota(int nfeatures, int nlevels, int* quota) {
  const float factor = (float)(1.0 / 1.2f);
  float ndesired = (float)(nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels)));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    quota[l] = cv_roundf(ndesired);
    sum += quota[l];
    ndesired *= factor;
  }
  quota[nlevels - 1] = std::max(nfeatures - sum, 0);
}

void orc_orb_resize(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) { resize_linear_u8(src, sw, sh, dst, dw, dh); }
int orc_orb_fast_score(const uint8_t* img, int w, int h, int x, int y, int thr) {
  if (x < 3 || y < 3 || x >= w - 3 || y >= h - 3) return 0;
  return fast_score(img, w, x, y, thr);
}
void orc_orb_gauss7(const uint8_t* src, int w, int h, uint8_t* dst) { gauss7(src, w, h, dst); }
float orc_orb_fast_atan2(float y, float x) { return fast_atan2(y, x); }

// kp: 5 floats per keypoint (x, y in level-0 pixels, angle in degrees, response, octave); desc: 32 bytes each.
int orc_orb_detect_describe(const uint8_t* img, int w, int h, size_t pitch, int nfeatures, float* kp, uint8_t* desc, int cap) {
  const int nlevels = 8, edge = 19, thr = 20, half_patch = 15;
  int lw[8], lh[8], quota[8];
  float scale[8];
  orc_orb_level_sizes(w, h, nlevels, lw, lh, scale);
  orc_orb_level_quota(nfeatures, nlevels, quota);
  std::vector<std::vector<uint8_t>> pyr(nlevels);
  pyr[0].resize((size_t)w * h);
  for (int y = 0; y < h; y++) std::memcpy(&pyr[0][(size_t)y * w], img + (size_t)y * pitch, w);
  for (int l = 1; l < nlevels; l++) {
    pyr[l].resize((size_t)lw[l] * lh[l]);
    resize_linear_u8(pyr[l - 1].data(), lw[l - 1], lh[l - 1], pyr[l].data(), lw[l], lh[l]);
  }
  int umax[16];
  {
    const int vmax = (int)std::floor(half_patch * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(half_patch * std::sqrt(2.0) / 2);
    for (int v = 0; v <= vmax; v++) umax[v] = cv_round(std::sqrt((double)half_patch * half_patch - v * v));
    for (int v = half_patch, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }
  int n_out = 0;
  for (int l = 0; l < nlevels; l++) {
    const int W = lw[l], H = lh[l];
    const uint8_t* im = pyr[l].data();
    if (W < 2 * edge + 1 || H < 2 * edge + 1) continue;
    std::vector<uint8_t> score((size_t)W * H, 0);
    for (int y = 3; y < H - 3; y++)
      for (int x = 3; x < W - 3; x++) score[(size_t)y * W + x] = (uint8_t)fast_score(im, W, x, y, thr);
    // non-maximum suppression (strict) + border filter, then the score histogram for retainBest
    std::vector<int> cand;  // y * W + x in raster order
    int hist[256] = {0};
    for (int y = edge; y < H - edge; y++)
      for (int x = edge; x < W - edge; x++) {
        const int s = score[(size_t)y * W + x];
        if (s == 0) continue;
        bool ok = true;
        for (int dy = -1; dy <= 1 && ok; dy++)
          for (int dx = -1; dx <= 1; dx++)
            if ((dx || dy) && !(s > score[(size_t)(y + dy) * W + x + dx])) {
              ok = false;
              break;
            }
        if (ok) {
          cand.push_back(y * W + x);
          hist[s]++;
        }
      }
    int cut = 0;  // keep score >= cut
    if ((int)cand.size() > quota[l]) {
      int acc = 0;
      for (cut = 255; cut > 0; cut--) {
        acc += hist[cut];
        if (acc >= quota[l]) break;
      }
    }
    if (quota[l] == 0) continue;
    std::vector<uint8_t> blurred((size_t)W * H);
    gauss7(im, W, H, blurred.data());
    for (int c : cand) {
      const int y = c / W, x = c % W, s = score[c];
      if (s < cut) continue;
      if (n_out >= cap) return n_out;
      // intensity centroid (orb.cpp ICAngles)
      int m_01 = 0, m_10 = 0;
      const uint8_t* center = im + (size_t)y * W + x;
      for (int u = -half_patch; u <= half_patch; ++u) m_10 += u * center[u];
      for (int v = 1; v <= half_patch; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
          const int val_plus = center[u + v * W], val_minus = center[u - v * W];
          v_sum += (val_plus - val_minus);
          m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
      }
      const float angle = fast_atan2((float)m_01, (float)m_10);
      float* k = kp + 5 * (size_t)n_out;
      k[0] = (float)x * scale[l];
      k[1] = (float)y * scale[l];
      k[2] = angle;
      k[3] = (float)s;
      k[4] = (float)l;
      const float rad = angle * (float)(M_PI / 180.0);
      const float a = (float)std::cos((double)rad), b = (float)std::sin((double)rad);
      const uint8_t* bc = blurred.data() + (size_t)y * W + x;
      uint8_t* dsc = desc + 32 * (size_t)n_out;
      for (int j = 0; j < 32; j++) {
        int byte = 0;
        for (int bit = 0; bit < 8; bit++) {
          const PatRow& p = kPattern[8 * j + bit];
          const float xa = (float)p.xa * a - (float)p.ya * b, ya = (float)p.xa * b + (float)p.ya * a;
          const float xb = (float)p.xb * a - (float)p.yb * b, yb = (float)p.xb * b + (float)p.yb * a;
          const int t0 = bc[cv_roundf(ya) * W + cv_roundf(xa)], t1 = bc[cv_roundf(yb) * W + cv_roundf(xb)];
          byte |= (t0 < t1) << bit;
        }
        dsc[j] = (uint8_t)byte;
      }
      n_out++;
    }
  }
  return n_out;
}

}  // extern "C"
