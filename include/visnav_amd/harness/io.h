// harness/io.h -- dataset and calibration input of the headless pipeline, restating the readers of the
// reference without their third-party dependencies:
//   * EuRoC image list        src/slam.cpp:1006-1040 (skip lines shorter than 20 chars or starting with
//                             '#', 19-digit ns timestamp, image name = line.substr(20, size - 21))
//   * ground truth            include/io/dataset_io_euroc.h:83-134 (state_groundtruth_estimate0/data.csv:
//                             t, p xyz, q wxyz, ...; or gt/data.csv)
//   * calibration             cereal JSON written by include/visnav/serialization.h:113-167
//                             ("value0": {"cam.T_i_c": [{px..qw}], "cam.intrinsics": [{cam_type, fx.., p1..p4, width, height}]})
//   * images                  pangolin::LoadImage stands behind the reference's loads; here: binary PGM (P5)
//                             and PNG (8-bit grey / RGB / RGBA / palette-free, non-interlaced) decoded with the
//                             inflate implementation below (RFC 1950/1951), no libpng / zlib needed.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../keypoints.h"  // mirror / reference types
#include "geometry.h"

namespace visnav {
namespace harness {

// ------------------------------------------------------------------------------------------ JSON
struct Json {
  enum Kind { Null, Num, Str, Arr, Obj, Bool } kind = Null;
  double num = 0;
  bool b = false;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json& at(const std::string& k) const {
    for (const auto& kv : obj)
      if (kv.first == k) return kv.second;
    throw std::runtime_error("json: missing key " + k);
  }
  bool has(const std::string& k) const {
    for (const auto& kv : obj)
      if (kv.first == k) return true;
    return false;
  }
};

class JsonParser {
 public:
  explicit JsonParser(const std::string& s) : s_(s) {}
  Json parse() {
    Json v = value();
    ws();
    if (i_ != s_.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& s_;
  size_t i_ = 0;
  [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(i_)); }
  void ws() {
    while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) i_++;
  }
  Json value() {
    ws();
    if (i_ >= s_.size()) fail("unexpected end");
    const char c = s_[i_];
    Json v;
    if (c == '{') {
      v.kind = Json::Obj;
      i_++;
      ws();
      if (i_ < s_.size() && s_[i_] == '}') {
        i_++;
        return v;
      }
      while (true) {
        ws();
        Json k = string_value();
        ws();
        if (i_ >= s_.size() || s_[i_] != ':') fail("expected ':'");
        i_++;
        v.obj.emplace_back(k.str, value());
        ws();
        if (i_ < s_.size() && s_[i_] == ',') {
          i_++;
          continue;
        }
        if (i_ < s_.size() && s_[i_] == '}') {
          i_++;
          return v;
        }
        fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      v.kind = Json::Arr;
      i_++;
      ws();
      if (i_ < s_.size() && s_[i_] == ']') {
        i_++;
        return v;
      }
      while (true) {
        v.arr.push_back(value());
        ws();
        if (i_ < s_.size() && s_[i_] == ',') {
          i_++;
          continue;
        }
        if (i_ < s_.size() && s_[i_] == ']') {
          i_++;
          return v;
        }
        fail("expected ',' or ']'");
      }
    }
    if (c == '"') return string_value();
    if (s_.compare(i_, 4, "true") == 0) {
      i_ += 4;
      v.kind = Json::Bool;
      v.b = true;
      return v;
    }
    if (s_.compare(i_, 5, "false") == 0) {
      i_ += 5;
      v.kind = Json::Bool;
      return v;
    }
    if (s_.compare(i_, 4, "null") == 0) {
      i_ += 4;
      return v;
    }
    char* end = nullptr;
    v.num = std::strtod(s_.c_str() + i_, &end);
    if (end == s_.c_str() + i_) fail("bad value");
    i_ = (size_t)(end - s_.c_str());
    v.kind = Json::Num;
    return v;
  }
  Json string_value() {
    if (i_ >= s_.size() || s_[i_] != '"') fail("expected string");
    i_++;
    Json v;
    v.kind = Json::Str;
    while (i_ < s_.size() && s_[i_] != '"') {
      if (s_[i_] == '\\' && i_ + 1 < s_.size()) {
        const char e = s_[i_ + 1];
        v.str.push_back(e == 'n' ? '\n' : (e == 't' ? '\t' : e));
        i_ += 2;
      } else {
        v.str.push_back(s_[i_++]);
      }
    }
    if (i_ >= s_.size()) fail("unterminated string");
    i_++;
    return v;
  }
};

inline std::string read_file(const std::string& path, bool* ok = nullptr) {
  std::ifstream f(path, std::ios::binary);
  if (ok) *ok = f.is_open();
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

// serialization.h:113-167
inline bool load_calibration(const std::string& path, Calibration& calib) {
  bool ok = false;
  const std::string txt = read_file(path, &ok);
  if (!ok) return false;
  const Json root = JsonParser(txt).parse();
  const Json& v = root.at("value0");
  calib.T_i_c.clear();
  calib.intrinsics.clear();
  for (const Json& t : v.at("cam.T_i_c").arr) {
    Sophus::SE3d T;
    double* d = T.data();
    d[0] = t.at("qx").num;
    d[1] = t.at("qy").num;
    d[2] = t.at("qz").num;
    d[3] = t.at("qw").num;
    d[4] = t.at("px").num;
    d[5] = t.at("py").num;
    d[6] = t.at("pz").num;
    calib.T_i_c.push_back(T);
  }
  for (const Json& c : v.at("cam.intrinsics").arr) {
    auto cam = std::make_shared<AbstractCameraD>();
    cam->model = c.at("cam_type").str;
    const char* keys[8] = {"fx", "fy", "cx", "cy", "p1", "p2", "p3", "p4"};
    for (int k = 0; k < 8; k++) cam->param[k] = c.at(keys[k]).num;
    cam->width_ = (int)c.at("width").num;
    cam->height_ = (int)c.at("height").num;
    calib.intrinsics.push_back(cam);
  }
  return calib.T_i_c.size() >= 2 && calib.intrinsics.size() >= 2;
}

// ------------------------------------------------------------------------------------------ inflate
class Inflater {
 public:
  Inflater(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  // zlib stream (RFC 1950): 2-byte header, deflate blocks, adler32 (not verified)
  bool run_zlib(std::vector<uint8_t>& out) {
    if (n_ < 2 || (p_[0] & 0x0F) != 8 || ((p_[0] << 8) | p_[1]) % 31 != 0 || (p_[1] & 0x20)) return false;
    pos_ = 2;
    return inflate(out);
  }

 private:
  const uint8_t* p_;
  size_t n_, pos_ = 0;
  uint32_t bitbuf_ = 0;
  int bitcnt_ = 0;
  bool err_ = false;
  uint32_t bits(int need) {
    uint32_t v = bitbuf_;
    while (bitcnt_ < need) {
      if (pos_ >= n_) {
        err_ = true;
        return 0;
      }
      v |= (uint32_t)p_[pos_++] << bitcnt_;
      bitcnt_ += 8;
    }
    bitbuf_ = need < 32 ? v >> need : 0;
    bitcnt_ -= need;
    return need < 32 ? v & ((1u << need) - 1) : v;
  }
  struct Huff {
    uint16_t count[16];
    uint16_t symbol[288];
  };
  static bool build(Huff& h, const uint8_t* len, int n) {
    for (int i = 0; i < 16; i++) h.count[i] = 0;
    for (int i = 0; i < n; i++) h.count[len[i]]++;
    if (h.count[0] == n) return true;
    int left = 1;
    for (int i = 1; i < 16; i++) {
      left <<= 1;
      left -= h.count[i];
      if (left < 0) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + h.count[i];
    for (int i = 0; i < n; i++)
      if (len[i]) h.symbol[offs[len[i]]++] = (uint16_t)i;
    return true;
  }
  int decode(const Huff& h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; len++) {
      code |= (int)bits(1);
      if (err_) return -1;
      const int count = h.count[len];
      if (code - count < first) return h.symbol[index + (code - first)];
      index += count;
      first += count;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
  bool codes(std::vector<uint8_t>& out, const Huff& lc, const Huff& dc) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    while (true) {
      int sym = decode(lc);
      if (sym < 0) return false;
      if (sym < 256) {
        out.push_back((uint8_t)sym);
      } else if (sym == 256) {
        return true;
      } else {
        sym -= 257;
        if (sym >= 29) return false;
        const int len = lbase[sym] + (int)bits(lext[sym]);
        const int ds = decode(dc);
        if (ds < 0 || ds >= 30) return false;
        const size_t dist = dbase[ds] + bits(dext[ds]);
        if (err_ || dist > out.size()) return false;
        const size_t start = out.size() - dist;
        for (int k = 0; k < len; k++) out.push_back(out[start + k]);
      }
    }
  }
  bool inflate(std::vector<uint8_t>& out) {
    int last;
    do {
      last = (int)bits(1);
      const int type = (int)bits(2);
      if (err_) return false;
      if (type == 0) {
        bitbuf_ = 0;
        bitcnt_ = 0;
        if (pos_ + 4 > n_) return false;
        const unsigned len = p_[pos_] | (p_[pos_ + 1] << 8), nlen = p_[pos_ + 2] | (p_[pos_ + 3] << 8);
        pos_ += 4;
        if ((len ^ 0xFFFFu) != nlen || pos_ + len > n_) return false;
        out.insert(out.end(), p_ + pos_, p_ + pos_ + len);
        pos_ += len;
      } else if (type == 1) {
        uint8_t l[320];
        int i = 0;
        for (; i < 144; i++) l[i] = 8;
        for (; i < 256; i++) l[i] = 9;
        for (; i < 280; i++) l[i] = 7;
        for (; i < 288; i++) l[i] = 8;
        Huff lc, dc;
        build(lc, l, 288);
        for (i = 0; i < 30; i++) l[i] = 5;
        build(dc, l, 30);
        if (!codes(out, lc, dc)) return false;
      } else if (type == 2) {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
        if (err_ || nlen > 286 || ndist > 30) return false;
        uint8_t l[320] = {0};
        for (int i = 0; i < ncode; i++) l[order[i]] = (uint8_t)bits(3);
        Huff cl;
        if (!build(cl, l, 19)) return false;
        uint8_t lens[320] = {0};
        int idx = 0;
        while (idx < nlen + ndist) {
          const int sym = decode(cl);
          if (sym < 0) return false;
          if (sym < 16) {
            lens[idx++] = (uint8_t)sym;
          } else {
            int prev = 0, rep;
            if (sym == 16) {
              if (idx == 0) return false;
              prev = lens[idx - 1];
              rep = 3 + (int)bits(2);
            } else if (sym == 17) {
              rep = 3 + (int)bits(3);
            } else {
              rep = 11 + (int)bits(7);
            }
            if (idx + rep > nlen + ndist) return false;
            while (rep--) lens[idx++] = (uint8_t)prev;
          }
        }
        Huff lc, dc;
        if (!build(lc, lens, nlen) || !build(dc, lens + nlen, ndist)) return false;
        if (!codes(out, lc, dc)) return false;
      } else {
        return false;
      }
    } while (!last);
    return !err_;
  }
};

// ------------------------------------------------------------------------------------------ images
struct GreyImage {
  int w = 0, h = 0;
  std::vector<uint8_t> px;
};

inline bool decode_pgm(const std::string& buf, GreyImage& img) {
  if (buf.size() < 2 || buf[0] != 'P' || buf[1] != '5') return false;
  size_t i = 2;
  auto next_int = [&](int& v) {
    while (i < buf.size()) {
      if (buf[i] == '#') {
        while (i < buf.size() && buf[i] != '\n') i++;
      } else if (buf[i] == ' ' || buf[i] == '\n' || buf[i] == '\t' || buf[i] == '\r') {
        i++;
      } else {
        break;
      }
    }
    if (i >= buf.size() || buf[i] < '0' || buf[i] > '9') return false;
    v = 0;
    while (i < buf.size() && buf[i] >= '0' && buf[i] <= '9') v = v * 10 + (buf[i++] - '0');
    return true;
  };
  int w, h, maxv;
  if (!next_int(w) || !next_int(h) || !next_int(maxv) || maxv != 255) return false;
  i++;  // single whitespace after maxval
  if (w <= 0 || h <= 0 || buf.size() < i + (size_t)w * h) return false;
  img.w = w;
  img.h = h;
  img.px.assign(buf.begin() + i, buf.begin() + i + (size_t)w * h);
  return true;
}

inline bool decode_png(const std::string& buf, GreyImage& img) {
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (buf.size() < 8 || std::memcmp(buf.data(), sig, 8) != 0) return false;
  const uint8_t* p = (const uint8_t*)buf.data();
  size_t i = 8;
  int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat;
  auto be32 = [&](size_t o) { return ((uint32_t)p[o] << 24) | ((uint32_t)p[o + 1] << 16) | ((uint32_t)p[o + 2] << 8) | p[o + 3]; };
  while (i + 12 <= buf.size()) {
    const uint32_t len = be32(i);
    const std::string type(buf, i + 4, 4);
    if (i + 12 + len > buf.size()) return false;
    if (type == "IHDR") {
      if (len < 13) return false;
      w = (int)be32(i + 8);
      h = (int)be32(i + 12);
      depth = p[i + 16];
      ctype = p[i + 17];
      interlace = p[i + 20];
    } else if (type == "IDAT") {
      idat.insert(idat.end(), p + i + 8, p + i + 8 + len);
    } else if (type == "IEND") {
      break;
    }
    i += 12 + len;
  }
  if (w <= 0 || h <= 0 || depth != 8 || interlace != 0) return false;
  int ch;
  if (ctype == 0) ch = 1;
  else if (ctype == 2) ch = 3;
  else if (ctype == 4) ch = 2;
  else if (ctype == 6) ch = 4;
  else return false;
  std::vector<uint8_t> raw;
  raw.reserve((size_t)h * ((size_t)w * ch + 1));
  if (!Inflater(idat.data(), idat.size()).run_zlib(raw)) return false;
  const size_t stride = (size_t)w * ch;
  if (raw.size() < (size_t)h * (stride + 1)) return false;
  std::vector<uint8_t> cur(stride), prev(stride, 0);
  img.w = w;
  img.h = h;
  img.px.resize((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* line = raw.data() + (size_t)y * (stride + 1);
    const int ft = line[0];
    for (size_t x = 0; x < stride; x++) {
      const int a = x >= (size_t)ch ? cur[x - ch] : 0, b = prev[x], c = x >= (size_t)ch ? prev[x - ch] : 0;
      int v = line[1 + x];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: {
          const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return false;
      }
      cur[x] = (uint8_t)v;
    }
    for (int x = 0; x < w; x++) {
      if (ch <= 2) {
        img.px[(size_t)y * w + x] = cur[(size_t)x * ch];
      } else {  // RGB(A) -> luma, integer BT.601 weights
        const uint8_t* q = &cur[(size_t)x * ch];
        img.px[(size_t)y * w + x] = (uint8_t)((299 * q[0] + 587 * q[1] + 114 * q[2] + 500) / 1000);
      }
    }
    prev.swap(cur);
  }
  return true;
}

inline bool load_image(const std::string& path, GreyImage& img) {
  bool ok = false;
  const std::string buf = read_file(path, &ok);
  if (!ok) return false;
  return decode_png(buf, img) || decode_pgm(buf, img);
}

// ------------------------------------------------------------------------------------------ EuRoC
struct EurocDataset {
  std::vector<int64_t> timestamps;                       // src/slam.cpp `timestamps`
  std::map<FrameCamId, std::string> images;              // src/slam.cpp `images`
  std::vector<int64_t> gt_t_ns;                          // ground truth (body frame)
  std::vector<Vec3> gt_t_w_i;
};

inline bool load_euroc(const std::string& dataset_path, EurocDataset& ds, int num_cams = 2) {
  std::ifstream times(dataset_path + "/cam0/data.csv");
  if (!times.is_open()) return false;
  int id = 0;
  std::string line;
  while (std::getline(times, line)) {
    if (line.size() < 20 || line[0] == '#') continue;
    ds.timestamps.push_back(std::strtoll(line.substr(0, 19).c_str(), nullptr, 10));
    const std::string img_name = line.substr(20, line.size() - 21);  // drops the trailing '\r' of the EuRoC files
    for (int c = 0; c < num_cams; c++)
      ds.images[FrameCamId(id, c)] = dataset_path + "/cam" + std::to_string(c) + "/data/" + img_name;
    id++;
  }
  auto read_gt = [&](const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::string l;
    while (std::getline(f, l)) {
      if (l.empty() || l[0] == '#') continue;
      std::stringstream ss(l);
      char tmp;
      uint64_t t;
      double px, py, pz;
      ss >> t >> tmp >> px >> tmp >> py >> tmp >> pz;
      if (ss.fail()) continue;
      ds.gt_t_ns.push_back((int64_t)t);
      ds.gt_t_w_i.emplace_back(px, py, pz);
    }
    return true;
  };
  if (!read_gt(dataset_path + "/state_groundtruth_estimate0/data.csv")) read_gt(dataset_path + "/gt/data.csv");
  return id > 0;
}

}  // namespace harness
}  // namespace visnav
