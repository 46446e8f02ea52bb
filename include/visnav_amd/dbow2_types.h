// visnav_amd/dbow2_types.h -- the two DBoW2 container types the reference's Camera carries
// (thirdparty/DBoW2_ORBSLAM/DBoW2/BowVector.h:55-57, FeatureVector.h:21-23): the same std::map types, so code that
// walks them (the inverted file, include/visnav/loop_closure_utils.h:156) is unchanged.
#pragma once
#include <map>
#include <vector>

namespace DBoW2 {
typedef unsigned int WordId;
typedef unsigned int NodeId;
typedef double WordValue;
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
}  // namespace DBoW2
