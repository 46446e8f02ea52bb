// ref_dbow2_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A driver around the REFERENCE's own DBoW2 container classes.  oracle/Makefile compiles it together with
// the reference's unmodified sources, where they lie,
//     /root/reference/thirdparty/DBoW2_ORBSLAM/DBoW2/BowVector.cpp      (addWeight :34-46, addIfNotExist :50-58,
//                                                                         normalize :62-84)
//     /root/reference/thirdparty/DBoW2_ORBSLAM/DBoW2/FeatureVector.cpp  (addFeature :30-44)
// into oracle/_ref/libdbow2_ref.so (git-ignored; never copied into the repository).  These two files are
// the only reference sources on the hot path that compile with the plain toolchain (everything else needs
// OpenCV / Eigen / Sophus / Ceres / Pangolin headers, which are empty submodules: no stand-ins are written).
// The functions below replay an operation stream on the reference classes and flatten the resulting
// std::map in iteration order, so tests can compare oracle/orc_bow.cpp's restatement bit for bit.
#include <cstdint>

#include "BowVector.h"
#include "FeatureVector.h"

extern "C" {

// ops[i]: 0 = addWeight(ids[i], vals[i]), 1 = addIfNotExist(ids[i], vals[i]).
// norm: 0 none, 1 normalize(L1), 2 normalize(L2).  Output capacity n.  Returns the number of entries.
int ref_bowvec_stream(const uint32_t* ids, const double* vals, const uint8_t* ops, int n, int norm, uint32_t* out_ids,
                      double* out_vals) {
  DBoW2::BowVector v;
  for (int i = 0; i < n; i++) {
    if (ops[i] == 0)
      v.addWeight(ids[i], vals[i]);
    else
      v.addIfNotExist(ids[i], vals[i]);
  }
  if (norm == 1) v.normalize(DBoW2::L1);
  if (norm == 2) v.normalize(DBoW2::L2);
  int j = 0;
  for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++j) {
    out_ids[j] = it->first;
    out_vals[j] = it->second;
  }
  return j;
}

// addFeature(nodes[i], feats[i]) for i = 0..n-1; output: (node, feature) pairs in map / vector order.
int ref_featvec_stream(const uint32_t* nodes, const uint32_t* feats, int n, uint32_t* out_nodes, uint32_t* out_feats) {
  DBoW2::FeatureVector fv;
  for (int i = 0; i < n; i++) fv.addFeature(nodes[i], feats[i]);
  int j = 0;
  for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it)
    for (size_t k = 0; k < it->second.size(); k++, j++) {
      out_nodes[j] = it->first;
      out_feats[j] = it->second[k];
    }
  return j;
}

}  // extern "C"
