// detect_loop_closure (include/visnav_amd/loop_closure.h; reference include/visnav/loop_closure_utils.h:294-388) on a
// constructed place-recognition scenario -- keyframes 0..9 along a path, keyframes 20.. revisiting the places of
// keyframes 3.. -- with BoW vectors in which a place has its own words.  Prints one line per new keyframe:
//   <frame> <returned> <n candidates> <first candidate> <groups> <max consistency>
// and the inverted-file population at the end.  The scores go through ORBVocabulary::score (the MI355X kernel).
#include <cstdio>
#include <vector>

#include "visnav_amd/loop_closure.h"

using namespace visnav;

static DBoW2::BowVector place_vector(int place, int variant) {
  DBoW2::BowVector v;
  for (int w = 0; w < 10; w++) v[(unsigned)w] = 1.0;                                 // words every image has
  for (int w = 0; w < 50; w++) v[(unsigned)(100 * (place + 1) + w)] = 1.0;            // the place's own words
  for (int w = 0; w < 10; w++) v[(unsigned)(5000 + 50 * variant + w)] = 1.0;          // view-specific clutter
  double s = 0;
  for (auto& kv : v) s += kv.second;
  for (auto& kv : v) kv.second /= s;
  return v;
}

int main() {
  ORBVocabularyAmd voc;  // score() needs no tree
  Cameras keyframes;
  CovisibilityGraph graph;
  DBoWInvertedFile db(20000);
  ConsistentGroups groups;
  std::vector<FrameCamId> enough;
  auto add = [&](int frame, int place, int variant, bool connect_prev, int prev_frame) {
    Camera cam;
    cam.bow_vector = place_vector(place, variant);
    const FrameCamId f(frame, 0);
    std::set<FrameCamId> edges;
    if (connect_prev) {
      const FrameCamId p(prev_frame, 0);
      cam.covisible_weights[p] = 40;
      edges.insert(p);
      graph[p].insert(f);
    }
    graph[f] = edges;
    const bool found = detect_loop_closure(f, cam, keyframes, db, &voc, graph, groups, enough, /*threshold*/ 20, /*num_consistency*/ 3);
    int maxc = 0;
    for (const auto& g : groups) maxc = g.second > maxc ? g.second : maxc;
    std::printf("%d %d %zu %lld %zu %d\n", frame, (int)found, enough.size(), enough.empty() ? -1LL : (long long)enough[0].frame_id, groups.size(), maxc);
    keyframes[f] = cam;
  };
  for (int i = 0; i < 10; i++) add(i, i, i, i > 0, i - 1);          // first pass: no place is seen twice
  add(20, 3, 100, true, 9);                                           // the revisit starts (covisible with its predecessor only)
  add(21, 4, 101, true, 20);
  add(22, 5, 102, true, 21);
  add(23, 6, 103, true, 22);                                          // fourth consistent detection: 0, 1, 2, 3 >= 3
  add(24, 30, 104, true, 23);                                         // a new place: no candidates, the groups are cleared
  size_t filled = 0, entries = 0;
  for (const auto& l : db) {
    filled += !l.empty();
    entries += l.size();
  }
  std::printf("db %zu %zu\n", filled, entries);
  return 0;
}
