// orc_bow.cpp -- CPU restatement of the DBoW2 pieces the reference's loop-closure scoring uses.
// TEST INFRASTRUCTURE ONLY (see vslam_oracle.h).
//
// Follows thirdparty/DBoW2_ORBSLAM/DBoW2/{TemplatedVocabulary.h, FORB.cpp, BowVector.cpp,
// FeatureVector.cpp, ScoringObject.cpp}; containers are the same std::map types the reference uses
// (BowVector = std::map<WordId,double>, FeatureVector = std::map<NodeId, std::vector<unsigned>>).
//
// Documented deviation: loadFromTextFile (TemplatedVocabulary.h:1380-1421) loops `while(!f.eof())`
// and therefore parses the empty string after the final newline as one more node: parent 0, not a
// leaf, weight 0, and a descriptor whose 32 bytes are never written (uninitialised cv::Mat memory,
// FORB.cpp:118-131).  That node becomes an 11th child of the root with undefined contents.  This
// restatement skips blank lines instead of reproducing undefined behaviour.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "vslam_oracle.h"

namespace {
struct Node {
  uint32_t id = 0;
  double weight = 0;
  std::vector<uint32_t> children;
  uint32_t parent = 0;
  uint8_t descriptor[32] = {0};
  uint32_t word_id = 0;
  bool isLeaf() const { return children.empty(); }
};

typedef std::map<uint32_t, double> BowVector;
typedef std::map<uint32_t, std::vector<uint32_t>> FeatureVector;

// FORB.cpp:81-101 (bit-count of the xor of two 256-bit strings)
int forb_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    std::memcpy(&pa, a + 4 * i, 4);
    std::memcpy(&pb, b + 4 * i, 4);
    uint32_t v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// BowVector::addWeight, BowVector.cpp:34-46
void bow_add_weight(BowVector& bow, uint32_t id, double v) {
  auto it = bow.lower_bound(id);
  if (it != bow.end() && !(bow.key_comp()(id, it->first)))
    it->second += v;
  else
    bow.insert(it, BowVector::value_type(id, v));
}

// BowVector::addIfNotExist, BowVector.cpp:50-58
void bow_add_if_not_exist(BowVector& bow, uint32_t id, double v) {
  auto it = bow.lower_bound(id);
  if (it == bow.end() || bow.key_comp()(id, it->first)) bow.insert(it, BowVector::value_type(id, v));
}

// BowVector::normalize, BowVector.cpp:62-84 (norm_l 1 = L1, 2 = L2)
void bow_normalize(BowVector& bow, int norm_l) {
  double norm = 0.0;
  if (norm_l == 1) {
    for (auto& kv : bow) norm += fabs(kv.second);
  } else {
    for (auto& kv : bow) norm += kv.second * kv.second;
    norm = sqrt(norm);
  }
  if (norm > 0.0)
    for (auto& kv : bow) kv.second /= norm;
}

// FeatureVector::addFeature, FeatureVector.cpp:30-44
void fv_add_feature(FeatureVector& fv, uint32_t id, uint32_t i_feature) {
  auto it = fv.lower_bound(id);
  if (it != fv.end() && it->first == id) {
    it->second.push_back(i_feature);
  } else {
    it = fv.insert(it, FeatureVector::value_type(id, std::vector<uint32_t>()));
    it->second.push_back(i_feature);
  }
}
}  // namespace

struct orc_voc {
  int k = 0, L = 0;
  int scoring = 0, weighting = 0;
  std::vector<Node> nodes;
  std::vector<uint32_t> words;  // word id -> node id
};

// TemplatedVocabulary.h:1338-1424
orc_voc* orc_voc_load_text(const char* path) {
  std::ifstream f(path);
  if (!f.is_open()) return nullptr;
  orc_voc* v = new orc_voc;
  std::string s;
  std::getline(f, s);
  {
    std::stringstream ss;
    ss << s;
    int n1 = -1, n2 = -1;
    ss >> v->k >> v->L >> n1 >> n2;
    if (ss.fail() || v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || n1 < 0 || n1 > 5 || n2 < 0 ||
        n2 > 3) {
      delete v;
      return nullptr;
    }
    v->scoring = n1;
    v->weighting = n2;
  }
  v->nodes.resize(1);
  v->nodes[0].id = 0;
  while (std::getline(f, s)) {
    if (s.find_first_not_of(" \t\r\n") == std::string::npos) continue;  // documented deviation
    std::stringstream ss;
    ss << s;
    const uint32_t nid = (uint32_t)v->nodes.size();
    v->nodes.resize(nid + 1);
    v->nodes[nid].id = nid;
    int pid = 0, leaf = 0;
    ss >> pid;
    v->nodes[nid].parent = (uint32_t)pid;
    v->nodes[pid].children.push_back(nid);
    ss >> leaf;
    for (int i = 0; i < 32; i++) {  // FORB::fromString, FORB.cpp:118-135
      int n = 0;
      ss >> n;
      if (!ss.fail()) v->nodes[nid].descriptor[i] = (uint8_t)n;
    }
    ss >> v->nodes[nid].weight;
    if (leaf > 0) {
      v->nodes[nid].word_id = (uint32_t)v->words.size();
      v->words.push_back(nid);
    }
  }
  return v;
}

void orc_voc_free(orc_voc* v) { delete v; }

void orc_voc_info(const orc_voc* v, int* k, int* L, int* n_nodes, int* n_words) {
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (n_nodes) *n_nodes = (int)v->nodes.size();
  if (n_words) *n_words = (int)v->words.size();
}

namespace {
// TemplatedVocabulary.h:1218-1259
void transform_one(const orc_voc* v, const uint8_t* feature, uint32_t& word_id, double& weight,
                   uint32_t* nid, int levelsup) {
  const int nid_level = v->L - levelsup;
  if (nid_level <= 0 && nid != nullptr) *nid = 0;
  uint32_t final_id = 0;
  int current_level = 0;
  do {
    ++current_level;
    const std::vector<uint32_t>& nodes = v->nodes[final_id].children;
    final_id = nodes[0];
    double best_d = forb_distance(feature, v->nodes[final_id].descriptor);
    for (size_t i = 1; i < nodes.size(); i++) {
      const uint32_t id = nodes[i];
      const double d = forb_distance(feature, v->nodes[id].descriptor);
      if (d < best_d) {
        best_d = d;
        final_id = id;
      }
    }
    if (nid != nullptr && current_level == nid_level) *nid = final_id;
  } while (!v->nodes[final_id].isLeaf());
  word_id = v->nodes[final_id].word_id;
  weight = v->nodes[final_id].weight;
}
}  // namespace

// TemplatedVocabulary.h:1127-1194 for weighting TF_IDF(0)/TF(1) and IDF(2)/BINARY(3); scoring L1
// variants normalise with the L1 norm (ScoringObject.h: mustNormalize -> L1), L2 with L2.
void orc_bow_transform(const orc_voc* v, const uint8_t* desc32, int n, int levelsup, uint32_t* word_ids,
                       double* word_vals, int* nnz, uint32_t* fv_node, uint32_t* fv_feat, int* fv_n) {
  BowVector bow;
  FeatureVector fv;
  *nnz = 0;
  *fv_n = 0;
  if (v->nodes.size() <= 1) return;  // empty()
  // scoring: 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT
  bool must = false;
  int norm_l = 1;
  switch (v->scoring) {
    case 0: must = true; norm_l = 1; break;
    case 1: must = true; norm_l = 2; break;
    case 2: must = true; norm_l = 1; break;
    case 3: must = true; norm_l = 1; break;
    case 4: must = true; norm_l = 1; break;
    default: must = false; break;
  }
  const bool tf = (v->weighting == 0 || v->weighting == 1);
  for (int i = 0; i < n; i++) {
    uint32_t id = 0, nid = 0;
    double w = 0;
    transform_one(v, desc32 + 32 * (size_t)i, id, w, &nid, levelsup);
    if (w > 0) {
      if (tf)
        bow_add_weight(bow, id, w);
      else
        bow_add_if_not_exist(bow, id, w);
      fv_add_feature(fv, nid, (uint32_t)i);
    }
  }
  if (tf && !bow.empty() && !must) {
    const double nd = (double)bow.size();
    for (auto& kv : bow) kv.second /= nd;
  }
  if (must) bow_normalize(bow, norm_l);
  int j = 0;
  for (auto& kv : bow) {
    word_ids[j] = kv.first;
    word_vals[j] = kv.second;
    j++;
  }
  *nnz = j;
  j = 0;
  for (auto& kv : fv)
    for (uint32_t f : kv.second) {
      fv_node[j] = kv.first;
      fv_feat[j] = f;
      j++;
    }
  *fv_n = j;
}

// ScoringObject.cpp:23-68
double orc_bow_score_l1(const uint32_t* ids1, const double* vals1, int n1, const uint32_t* ids2,
                        const double* vals2, int n2) {
  BowVector v1, v2;
  for (int i = 0; i < n1; i++) v1[ids1[i]] = vals1[i];
  for (int i = 0; i < n2; i++) v2[ids2[i]] = vals2[i];
  auto v1_it = v1.begin(), v2_it = v2.begin();
  const auto v1_end = v1.end(), v2_end = v2.end();
  double score = 0;
  while (v1_it != v1_end && v2_it != v2_end) {
    const double vi = v1_it->second, wi = v2_it->second;
    if (v1_it->first == v2_it->first) {
      score += fabs(vi - wi) - fabs(vi) - fabs(wi);
      ++v1_it;
      ++v2_it;
    } else if (v1_it->first < v2_it->first) {
      v1_it = v1.lower_bound(v2_it->first);
    } else {
      v2_it = v2.lower_bound(v1_it->first);
    }
  }
  score = -score / 2.0;
  return score;
}

// Operation-stream replays of the container restatements above; oracle/ref_dbow2_driver.cpp replays the same
// streams on the reference's own DBoW2::BowVector / FeatureVector objects (oracle/_ref/libdbow2_ref.so) and
// tests/test_oracle_ref_pin.py asserts identical keys, value bit patterns and order.
int orc_bowvec_stream(const uint32_t* ids, const double* vals, const uint8_t* ops, int n, int norm, uint32_t* out_ids,
                      double* out_vals) {
  BowVector bow;
  for (int i = 0; i < n; i++) {
    if (ops[i] == 0)
      bow_add_weight(bow, ids[i], vals[i]);
    else
      bow_add_if_not_exist(bow, ids[i], vals[i]);
  }
  if (norm == 1 || norm == 2) bow_normalize(bow, norm);
  int j = 0;
  for (auto& kv : bow) {
    out_ids[j] = kv.first;
    out_vals[j] = kv.second;
    j++;
  }
  return j;
}

int orc_featvec_stream(const uint32_t* nodes, const uint32_t* feats, int n, uint32_t* out_nodes, uint32_t* out_feats) {
  FeatureVector fv;
  for (int i = 0; i < n; i++) fv_add_feature(fv, nodes[i], feats[i]);
  int j = 0;
  for (auto& kv : fv)
    for (uint32_t f : kv.second) {
      out_nodes[j] = kv.first;
      out_feats[j] = f;
      j++;
    }
  return j;
}
