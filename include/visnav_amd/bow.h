// visnav_amd/bow.h -- drop-in for the DBoW2 calls the reference makes when scoring loop-closure
// candidates: ORBVocabulary::loadFromTextFile / transform / score
// (thirdparty/DBoW2_ORBSLAM/DBoW2/TemplatedVocabulary.h:1338, :1127, :1199; call sites
// include/visnav/keypoints.h:253, include/visnav/loop_closure_utils.h:119, :201, include/visnav/tracking.h:208).
// BowVector / FeatureVector keep DBoW2's container types (std::map), so the callers' code that walks
// them (inverted file at loop_closure_utils.h:156) is unchanged.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "dbow2_types.h"
#include "keypoints.h"

namespace visnav {

class ORBVocabularyAmd {
 public:
  ORBVocabularyAmd() = default;
  ORBVocabularyAmd(const ORBVocabularyAmd&) = delete;
  ~ORBVocabularyAmd() { vsl_voc_destroy(voc_); }

  // TemplatedVocabulary.h:1338: returns false on failure like the original
  bool loadFromTextFile(const std::string& filename) {
    vsl_voc_destroy(voc_);
    voc_ = nullptr;
    return vsl_voc_load_text(amd::ctx(), filename.c_str(), &voc_) == VSL_OK;
  }
  unsigned int size() const {
    int w = 0;
    if (voc_) vsl_voc_info(voc_, nullptr, nullptr, nullptr, &w);
    return (unsigned)w;
  }
  bool empty() const { return size() == 0; }

  // transform(features, v, fv, levelsup): features are 32-byte rows in cv::Mat / DBoW2 byte order.
  void transform(const std::vector<const uint8_t*>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv,
                 int levelsup) const {
    std::vector<uint8_t> flat(32 * features.size());
    for (size_t i = 0; i < features.size(); i++) std::memcpy(&flat[32 * i], features[i], 32);
    transform(flat.data(), (int)features.size(), v, fv, levelsup);
  }
  void transform(const uint8_t* desc32, int n, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
    v.clear();
    fv.clear();
    if (!voc_ || n <= 0) return;
    std::vector<uint32_t> ids(n), fn(n), ff(n);
    std::vector<double> vals(n);
    int nnz = 0, fvn = 0;
    amd::check(vsl_bow_transform(amd::ctx(), voc_, desc32, n, levelsup, ids.data(), vals.data(), &nnz, fn.data(),
                                 ff.data(), &fvn), "ORBVocabulary::transform");
    for (int i = 0; i < nnz; i++) v.emplace_hint(v.end(), ids[i], vals[i]);
    for (int i = 0; i < fvn; i++) fv[fn[i]].push_back(ff[i]);
  }
  // descriptors straight from KeypointsData (bitset<256>), converted with converter.h's bit order
  void transform(const std::vector<std::bitset<256>>& desc, DBoW2::BowVector& v, DBoW2::FeatureVector& fv,
                 int levelsup) const {
    std::vector<uint8_t> flat(32 * desc.size());
    vsl_desc_bitset_to_bytes(reinterpret_cast<const uint64_t*>(desc.data()), (int)desc.size(), flat.data());
    transform(flat.data(), (int)desc.size(), v, fv, levelsup);
  }

  // compute_bow_vector of include/visnav/keypoints.h:243-254 in one device pass: the ORB front end
  // (cv::ORB::create(num_features, 1.2, 8, 19, 0, 2, FAST_SCORE)->detectAndCompute, restated -- see
  // oracle/orc_orb.cpp for the arithmetic conventions) followed by transform(..., levelsup = 4)
  void compute_bow_vector(const pangolin::ManagedImage<uint8_t>& img_raw, int num_features, DBoW2::BowVector& v,
                          DBoW2::FeatureVector& fv, int levelsup = 4) const {
    v.clear();
    fv.clear();
    if (!voc_) return;
    const int cap = 2 * num_features + 512;
    std::vector<uint32_t> ids(cap), fn(cap), ff(cap);
    std::vector<double> vals(cap);
    int nnz = 0, fvn = 0;
    amd::check(vsl_compute_bow_vector(amd::ctx(), voc_, img_raw.ptr, (int)img_raw.w, (int)img_raw.h, img_raw.pitch, num_features,
                                      levelsup, cap, ids.data(), vals.data(), &nnz, fn.data(), ff.data(), &fvn),
               "compute_bow_vector");
    for (int i = 0; i < nnz; i++) v.emplace_hint(v.end(), ids[i], vals[i]);
    for (int i = 0; i < fvn; i++) fv[fn[i]].push_back(ff[i]);
  }

  // TemplatedVocabulary.h:1199-1203
  double score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const {
    std::vector<const DBoW2::BowVector*> one(1, &b);
    return score_batch(a, one)[0];
  }
  // one query against many candidates in ONE launch (what detect_loop_candidates does in a loop,
  // include/visnav/loop_closure_utils.h:199-211)
  std::vector<double> score_batch(const DBoW2::BowVector& q, const std::vector<const DBoW2::BowVector*>& cands) const {
    std::vector<uint32_t> qi, ci;
    std::vector<double> qv, cv;
    std::vector<int32_t> off(1, 0);
    for (const auto& kv : q) { qi.push_back(kv.first); qv.push_back(kv.second); }
    for (const auto* c : cands) {
      for (const auto& kv : *c) { ci.push_back(kv.first); cv.push_back(kv.second); }
      off.push_back((int32_t)ci.size());
    }
    std::vector<double> s(cands.size(), 0.0);
    if (cands.empty()) return s;
    amd::check(vsl_bow_score_batch(amd::ctx(), qi.data(), qv.data(), (int)qi.size(), ci.data(), cv.data(), off.data(),
                                   (int)cands.size(), s.data()), "ORBVocabulary::score");
    return s;
  }

 private:
  vsl_voc* voc_ = nullptr;
};

// include/visnav/keypoints.h:243-254 with the reference's argument order; the cv::Ptr<cv::ORB> argument of the
// reference is re-created inside it on every call with fixed parameters, so it carries no state and is dropped.
inline void compute_bow_vector(const pangolin::ManagedImage<uint8_t>& img_raw, int num_features, const ORBVocabularyAmd* voc,
                               DBoW2::BowVector& bow_vector, DBoW2::FeatureVector& feature_vector) {
  voc->compute_bow_vector(img_raw, num_features, bow_vector, feature_vector, 4);
}

}  // namespace visnav
