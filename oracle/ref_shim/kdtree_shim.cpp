// kdtree_shim.cpp -- C entry point over the REFERENCE's own kd-tree, compiled from the sources where they lie
// (/root/reference/src/vslam/feature/kdtree/KDTree.{hpp,cpp}, STL only) into oracle/_ref/libref_kdtree.so by
// oracle/Makefile.  Test infrastructure: it pins the nearest-neighbour rejection rule of the oracle
// and of select_kernel against the reference's code.  Nothing of the reference is copied into this repository;
// this file only restates the CALL SITE FeatureDetectorShiTomasiBucketing.cpp:127-130,168-185.
#include <cmath>
#include <cstdint>
#include <vector>

#include "KDTree.hpp"

using vslam::feature::utility::KDTree;
using vslam::feature::utility::point_t;
using vslam::feature::utility::pointVec;

extern "C" {

// pts: n existing features (x, y); q: nq candidate corners.  accept[i] = what :172-185 decides for candidate i
// (tree empty -> accept; else accept iff ||q - nearest|| > min_dist); nn[2i..] = the nearest point returned
// (NaN when the tree is empty).
int ref_kdtree_accept(const double* pts, int n, const double* q, int nq, double min_dist, uint8_t* accept,
                      double* nn) {
  pointVec kd_points;
  for (int i = 0; i < n; i++) kd_points.push_back(point_t{pts[2 * i], pts[2 * i + 1]});  // :127
  KDTree kd_tree(kd_points);                                                              // :130
  for (int i = 0; i < nq; i++) {
    point_t check_pt{q[2 * i], q[2 * i + 1]};
    auto nnkp = kd_tree.nearest_point(check_pt);  // :170
    if (nnkp.empty()) {                           // :172
      accept[i] = 1;
      nn[2 * i] = nn[2 * i + 1] = NAN;
    } else {
      const double dx = check_pt[0] - nnkp[0], dy = check_pt[1] - nnkp[1];
      accept[i] = std::sqrt(dx * dx + dy * dy) > min_dist ? 1 : 0;  // :178-180 Eigen (a - b).norm() > min
      nn[2 * i] = nnkp[0];
      nn[2 * i + 1] = nnkp[1];
    }
  }
  return 0;
}
}
