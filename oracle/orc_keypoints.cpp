// orc_keypoints.cpp -- CPU restatement of include/visnav/keypoints.h.  TEST INFRASTRUCTURE ONLY
// (see vslam_oracle.h for who may use it and for the parity-pinning status).
//
// Build with -ffp-contract=off: the fp32 detector stages are defined as individually rounded
// IEEE operations so that the HIP kernels can reproduce them bit for bit.
#include <algorithm>
#include <bitset>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "vslam_oracle.h"

namespace {

struct PatternRow {
  signed char xa, ya, xb, yb;
};
// keypoints.h:55-131 (numbers only; see tools/gen_pattern.py)
const PatternRow kPattern[256] = {
#include "rbrief_pattern.inc"
};

const int HALF_PATCH_SIZE = 15;  // keypoints.h:50
const int EDGE_THRESHOLD = 19;   // keypoints.h:51

// cv::BORDER_REFLECT_101 for a one-pixel overshoot (all this path needs).
inline int reflect101(int i, int n) {
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

struct Img {
  const uint8_t* p;
  int w, h;
  size_t pitch;
  // pangolin::Image::operator()(x, y): unchecked row-major access.
  inline uint8_t operator()(int x, int y) const { return p[(size_t)y * pitch + x]; }
  inline uint8_t refl(int x, int y) const { return (*this)(reflect101(x, w), reflect101(y, h)); }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// [upstream] cv::cornerMinEigenVal(src 8U, blockSize 3, ksize 3, BORDER_DEFAULT), OpenCV 4.x
// modules/imgproc/src/corner.cpp cornerEigenValsVecs + calcMinEigenVal, scalar (non-SIMD, no-FMA)
// paths, restated as:
//   scale = 1 / (2^(ksize-1) * blockSize * 255) = 1/3060, s = (float)scale
//   Dx = sepFilter(row [-1 0 1], col [s 2s s]):   Rx = I(x+1) - I(x-1);  Dx = (Rx(y-1) + Rx(y+1))*s + Rx(y)*2s
//   Dy = sepFilter(row [s 2s s], col [-1 0 1]):   Ry = ((s*I(x-1)) + (2s*I(x))) + (s*I(x+1)); Dy = Ry(y+1) - Ry(y-1)
//   cov = (Dx*Dx, Dx*Dy, Dy*Dy) in fp32
//   3x3 unnormalised box sum, accumulated in double (boxFilter uses a CV_64F sum buffer for 32F
//   sources) as row sums then column sums in a FIXED order (OpenCV's running-sum order depends on
//   the whole row/column history and is not reproducible in parallel; the sums are exact in double
//   unless a gradient is ~1e-10, so the order only matters in those cases), rounded to fp32 once;
//   BORDER_REFLECT_101 applied to the cov image
//   a = A*0.5f, b = B, c = C*0.5f;  lambda_min = (a + c) - sqrtf((a - c)*(a - c) + b*b)   (fp32, no FMA)
void orc_min_eig_response(const uint8_t* img, int w, int h, size_t pitch, float* resp) {
  Img I{img, w, h, pitch};
  const float s = (float)(1.0 / (4.0 * 3.0 * 255.0));
  const float s2 = 2.0f * s;
  std::vector<float> cxx((size_t)w * h), cxy((size_t)w * h), cyy((size_t)w * h);
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      float rx[3], ry[3];
      for (int k = -1; k <= 1; k++) {
        const int yy = y + k;
        const float l = (float)I.refl(x - 1, yy), m = (float)I.refl(x, yy), r = (float)I.refl(x + 1, yy);
        rx[k + 1] = r - l;
        float t = s * l;
        t = t + s2 * m;
        t = t + s * r;
        ry[k + 1] = t;
      }
      const float dx = (rx[0] + rx[2]) * s + rx[1] * s2;
      const float dy = ry[2] - ry[0];
      cxx[(size_t)y * w + x] = dx * dx;
      cxy[(size_t)y * w + x] = dx * dy;
      cyy[(size_t)y * w + x] = dy * dy;
    }
  }
  // row sums R(x,y) = (c(x-1,y) + c(x,y)) + c(x+1,y) in double, then A = (R(x,y-1) + R(x,y)) + R(x,y+1)
  std::vector<double> rxx((size_t)w * h), rxy((size_t)w * h), ryy((size_t)w * h);
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      const size_t l = (size_t)y * w + reflect101(x - 1, w), m = (size_t)y * w + x, r = (size_t)y * w + reflect101(x + 1, w);
      rxx[m] = ((double)cxx[l] + (double)cxx[m]) + (double)cxx[r];
      rxy[m] = ((double)cxy[l] + (double)cxy[m]) + (double)cxy[r];
      ryy[m] = ((double)cyy[l] + (double)cyy[m]) + (double)cyy[r];
    }
  }
  for (int y = 0; y < h; y++) {
    const size_t u = (size_t)reflect101(y - 1, h) * w, m = (size_t)y * w, d = (size_t)reflect101(y + 1, h) * w;
    for (int x = 0; x < w; x++) {
      const double A = (rxx[u + x] + rxx[m + x]) + rxx[d + x];
      const double B = (rxy[u + x] + rxy[m + x]) + rxy[d + x];
      const double C = (ryy[u + x] + ryy[m + x]) + ryy[d + x];
      const float a = (float)A * 0.5f, b = (float)B, c = (float)C * 0.5f;
      const float dd = a - c;
      float t = dd * dd;
      const float bb = b * b;
      t = t + bb;
      resp[(size_t)y * w + x] = (a + c) - sqrtf(t);
    }
  }
}

// [upstream] cv::goodFeaturesToTrack (modules/imgproc/src/featureselect.cpp), no mask, no Harris:
//   maxVal = max(eig); thr = maxVal*quality; eig = eig > thr ? eig : 0 (THRESH_TOZERO)
//   candidates: 1 <= x < w-1, 1 <= y < h-1, eig != 0 && eig == 3x3 dilate(eig)
//   sort descending by value, equal values by address descending (greaterThanPtr)
//   greedy: grid of cell_size = cvRound(min_dist); accept unless an accepted corner in the 3x3
//   neighbouring cells has dx*dx + dy*dy < min_dist^2; stop at max_corners.
int orc_good_features(const uint8_t* img, int w, int h, size_t pitch, int max_corners, double quality,
                      double min_dist, int* xy, float* response_out) {
  std::vector<float> eig((size_t)w * h);
  orc_min_eig_response(img, w, h, pitch, eig.data());
  if (response_out) std::memcpy(response_out, eig.data(), sizeof(float) * eig.size());
  double maxVal = 0;
  {
    // cv::minMaxLoc: true maximum (first occurrence); starts from the first element.
    float m = eig[0];
    for (size_t i = 1; i < eig.size(); i++) m = eig[i] > m ? eig[i] : m;
    maxVal = m;
  }
  const float thr = (float)(maxVal * quality);
  for (auto& v : eig) v = v > thr ? v : 0.0f;
  std::vector<const float*> cand;
  for (int y = 1; y < h - 1; y++) {
    for (int x = 1; x < w - 1; x++) {
      const float val = eig[(size_t)y * w + x];
      if (val == 0) continue;
      float mx = val;  // dilate with a 3x3 rectangle
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) mx = std::max(mx, eig[(size_t)(y + dy) * w + (x + dx)]);
      if (val == mx) cand.push_back(&eig[(size_t)y * w + x]);
    }
  }
  if (cand.empty()) return 0;
  std::sort(cand.begin(), cand.end(), [](const float* a, const float* b) {
    return (*a > *b) ? true : (*a < *b) ? false : (a > b);
  });
  int ncorners = 0;
  if (min_dist >= 1) {
    const int cell = (int)std::lrint(min_dist);  // cvRound
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    std::vector<std::vector<std::pair<float, float>>> grid((size_t)gw * gh);
    const double md2 = min_dist * min_dist;
    for (size_t i = 0; i < cand.size(); i++) {
      const int ofs = (int)(cand[i] - eig.data());
      const int y = ofs / w, x = ofs - y * w;
      bool good = true;
      const int xc = x / cell, yc = y / cell;
      const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
      const int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
      for (int yy = y1; yy <= y2 && good; yy++)
        for (int xx = x1; xx <= x2 && good; xx++)
          for (const auto& m : grid[(size_t)yy * gw + xx]) {
            const float dx = x - m.first, dy = y - m.second;
            if (dx * dx + dy * dy < md2) {
              good = false;
              break;
            }
          }
      if (good) {
        grid[(size_t)yc * gw + xc].emplace_back((float)x, (float)y);
        xy[2 * ncorners] = x;
        xy[2 * ncorners + 1] = y;
        ++ncorners;
        if (max_corners > 0 && ncorners == max_corners) break;
      }
    }
  } else {
    for (size_t i = 0; i < cand.size(); i++) {
      const int ofs = (int)(cand[i] - eig.data());
      xy[2 * ncorners] = ofs % w;
      xy[2 * ncorners + 1] = ofs / w;
      ++ncorners;
      if (max_corners > 0 && ncorners == max_corners) break;
    }
  }
  return ncorners;
}

// keypoints.h:133-150
int orc_detect_keypoints(const uint8_t* img, int w, int h, size_t pitch, int num_features,
                         double* corners_xy) {
  std::vector<int> pts(2 * (size_t)std::max(num_features, 1));
  const int n = orc_good_features(img, w, h, pitch, num_features, 0.01, 8, pts.data(), nullptr);
  int m = 0;
  for (int i = 0; i < n; i++) {
    const float x = (float)pts[2 * i], y = (float)pts[2 * i + 1];
    // pangolin::Image::InBounds(float x, float y, float border)
    const float border = (float)EDGE_THRESHOLD;
    if (border <= x && x < (float)(w - EDGE_THRESHOLD) && border <= y && y < (float)(h - EDGE_THRESHOLD)) {
      corners_xy[2 * m] = x;
      corners_xy[2 * m + 1] = y;
      m++;
    }
  }
  return m;
}

// keypoints.h:152-189
void orc_compute_angles(const uint8_t* img, int w, int h, size_t pitch, const double* corners_xy,
                        int n, int rotate_features, double* angles) {
  Img I{img, w, h, pitch};
  for (int i = 0; i < n; i++) {
    const int cx = (int)corners_xy[2 * i];
    const int cy = (int)corners_xy[2 * i + 1];
    double angle = 0;
    if (rotate_features) {
      double m01 = 0, m10 = 0;
      for (int x = -HALF_PATCH_SIZE; x <= HALF_PATCH_SIZE; x++) {
        for (int y = -HALF_PATCH_SIZE; y <= HALF_PATCH_SIZE; y++) {
          if (x * x + y * y <= HALF_PATCH_SIZE * HALF_PATCH_SIZE) {
            m01 += y * I(x + cx, y + cy);
            m10 += x * I(x + cx, y + cy);
          }
        }
      }
      angle = atan2(m01, m10);
    }
    angles[i] = angle;
  }
}

void orc_patch_moments(const uint8_t* img, int w, int h, size_t pitch, const double* corners_xy,
                       int n, int64_t* m01o, int64_t* m10o) {
  Img I{img, w, h, pitch};
  for (int i = 0; i < n; i++) {
    const int cx = (int)corners_xy[2 * i], cy = (int)corners_xy[2 * i + 1];
    int64_t m01 = 0, m10 = 0;
    for (int x = -HALF_PATCH_SIZE; x <= HALF_PATCH_SIZE; x++)
      for (int y = -HALF_PATCH_SIZE; y <= HALF_PATCH_SIZE; y++)
        if (x * x + y * y <= HALF_PATCH_SIZE * HALF_PATCH_SIZE) {
          m01 += y * I(x + cx, y + cy);
          m10 += x * I(x + cx, y + cy);
        }
    m01o[i] = m01;
    m10o[i] = m10;
  }
}

// keypoints.h:191-221.  cos/sin are evaluated per use, like the reference (8 libm calls per bit).
void orc_compute_descriptors(const uint8_t* img, int w, int h, size_t pitch, const double* corners_xy,
                             const double* angles, int n, uint64_t* desc) {
  Img I{img, w, h, pitch};
  for (int k = 0; k < n; k++) {
    std::bitset<256> descriptor;
    const double angle = angles[k];
    const int cx = (int)corners_xy[2 * k];
    const int cy = (int)corners_xy[2 * k + 1];
    for (int i = 0; i < 256; i++) {
      const PatternRow& p = kPattern[i];
      int xa = (int)round(cos(angle) * p.xa - sin(angle) * p.ya);
      int ya = (int)round(sin(angle) * p.xa + cos(angle) * p.ya);
      int xb = (int)round(cos(angle) * p.xb - sin(angle) * p.yb);
      int yb = (int)round(sin(angle) * p.xb + cos(angle) * p.yb);
      descriptor.set(i, I(xa + cx, ya + cy) < I(xb + cx, yb + cy));
    }
    static_assert(sizeof(std::bitset<256>) == 32, "bitset<256> must be 4 x u64");
    std::memcpy(desc + 4 * (size_t)k, &descriptor, 32);
  }
}

// keypoints.h:223-229
int orc_detect_describe(const uint8_t* img, int w, int h, size_t pitch, int num_features,
                        int rotate_features, double* corners_xy, double* angles, uint64_t* desc) {
  const int n = orc_detect_keypoints(img, w, h, pitch, num_features, corners_xy);
  orc_compute_angles(img, w, h, pitch, corners_xy, n, rotate_features, angles);
  orc_compute_descriptors(img, w, h, pitch, corners_xy, angles, n, desc);
  return n;
}

namespace {
typedef std::bitset<256> Desc;

// keypoints.h:278-313
bool isPQiffQP(const std::vector<Desc>& d1, const std::vector<Desc>& d2, int cd1_id, int cd2_id,
               int threshold, double dist_2_best) {
  int id1 = 0, best1_id = 0, best1_d = 256, best2_d = 256;
  const Desc& cd2 = d2[cd2_id];
  for (const auto& cd1 : d1) {
    const int d = (int)(cd1 ^ cd2).count();
    if (d < best2_d) {
      if (d < best1_d) {
        best2_d = best1_d;
        best1_d = d;
        best1_id = id1;
      } else {
        best2_d = d;
      }
    }
    id1++;
  }
  if (best1_d >= threshold) return false;
  if (best2_d < best1_d * dist_2_best) return false;
  return cd1_id == best1_id;
}
}  // namespace

// keypoints.h:323-369
int orc_match_descriptors(const uint64_t* d1p, int n1, const uint64_t* d2p, int n2, int threshold,
                          double dist_2_best, int32_t* pairs) {
  std::vector<Desc> d1(n1), d2(n2);
  if (n1) std::memcpy((void*)d1.data(), d1p, 32 * (size_t)n1);
  if (n2) std::memcpy((void*)d2.data(), d2p, 32 * (size_t)n2);
  int nm = 0, id1 = 0;
  for (const auto& cd1 : d1) {
    int id2 = 0, best1_id = 0, best1_d = 256, best2_d = 256;
    for (const auto& cd2 : d2) {
      const int d = (int)(cd1 ^ cd2).count();
      if (d < best2_d) {
        if (d < best1_d) {
          best2_d = best1_d;
          best1_d = d;
          best1_id = id2;
        } else {
          best2_d = d;
        }
      }
      id2++;
    }
    if (best1_d >= threshold) {
    } else if (best2_d < best1_d * dist_2_best) {
    } else if (n2 > 0 && isPQiffQP(d1, d2, id1, best1_id, threshold, dist_2_best)) {
      pairs[2 * nm] = id1;
      pairs[2 * nm + 1] = best1_id;
      nm++;
    }
    id1++;
  }
  return nm;
}

// converter.h:23-33
void orc_bitset_to_bytes(const uint64_t* desc, uint8_t* out32) {
  Desc d;
  std::memcpy((void*)&d, desc, 32);
  std::memset(out32, 0, 32);
  uint8_t* p = out32;
  for (size_t i = 0; i < 256; i++) {
    if (d.test(i)) *p |= 1 << (7 - (i % 8));
    if (i % 8 == 7) p++;
  }
}

// converter.h:50-61
void orc_bytes_to_bitset(const uint8_t* in32, uint64_t* desc) {
  Desc d;
  const uint8_t* p = in32;
  for (size_t i = 0; i < 256; i++) {
    std::bitset<8> temp(*p);
    if (temp.test(7 - (i % 8))) d.set(i);
    if (i % 8 == 7) p++;
  }
  std::memcpy(desc, &d, 32);
}
