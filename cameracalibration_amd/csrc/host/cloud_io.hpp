// cloud_io.hpp -- host-side ingest of the optimiser's inputs.
//
// Restates, without PCL/Boost, what Optimizer::getInputs and its helpers read:
//   readPointCloud            src/helpers.cpp:108-207   (.pcd / .ply / .xyz, metric scaling)
//   readTransformations       src/helpers.cpp:209-261   (affine branch: x,y,z,rz,ry,rx per line)
//   getTransVector            src/camera_calibration_optimization.cpp:70-98
//   getFileId                 src/camera_calibration_optimization.cpp:131-146
//   z <= 0.8 filter           src/camera_calibration_optimization.cpp:167-169
//   VoxelGrid leaf 0.02       src/camera_calibration_optimization.cpp:196-205 (PCL stand-in)
#pragma once

#include <cstddef>
#include <string>
#include <vector>

namespace ccal_host {

// AoS float xyz, stride 3
using Cloud = std::vector<float>;

double metric_scale(const std::string& metric);  // "mm" -> 1000, "cm" -> 100, else 1

// Throws std::runtime_error on unreadable / unsupported files (the reference
// terminates with a bare `throw;` there).
Cloud read_point_cloud(const std::string& filename, const std::string& metric);

Cloud read_ply(const std::string& filename);
Cloud read_pcd(const std::string& filename);
Cloud read_xyz_text(const std::string& filename, double scale);

// keeps points with z <= z_max (NaN z is dropped, as `pts.z<=0.8` does).  The comparison is in
// double as in the reference, where the float z is promoted against the literal 0.8: a point
// with z == 0.8f (0.800000011920929, e.g. 800 mm read with the mm metric) is dropped.
Cloud filter_z(const Cloud& in, double z_max);

// Deterministic voxel-grid centroid filter: PCL's index arithmetic (float
// inverse leaf, floor(p*inv) - min_b, x-fastest linear index), voxels emitted in
// ascending index, points inside a voxel accumulated in input order in float.
Cloud voxel_grid(const Cloud& in, float leaf);

// "a,b,c,d,e,f" -> 6 doubles via stof; first three divided by scale; angles
// through the int-truncating degree helper when angle_is_degree.  Returns the
// number of fields parsed.
int parse_pose_csv(const std::string& line, double scale, bool angle_is_degree, double pose[6]);

// One pose per non-empty line until the first empty line; translations / scale.
std::vector<std::vector<double>> read_transformations(const std::string& filename,
                                                      const std::string& metric);

int file_id(const std::string& filename);

} // namespace ccal_host
