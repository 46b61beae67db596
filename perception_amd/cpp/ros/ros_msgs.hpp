// ros_msgs.hpp - what pcl::toROSMsg(const pcl::PointCloud<pcl::PointXYZ>&, sensor_msgs::PointCloud2&) puts on the wire
// (icp.cpp:125,193-194; opd.cpp:207,242,259): the point structs as they lie in memory - 16-byte records x, y, z, 1.0f -,
// fields x / y / z float32 at offsets 0 / 4 / 8, height 1, is_dense, little endian, the cloud's header.
#pragma once
#include <cstring>
#include <sensor_msgs/PointCloud2.h>

#include "../pcl_compat.hpp"

namespace pclhip {
// cloud == nullptr: n records x = y = z = 0 (to be filled in place by cd_get_cluster_points with stride 16)
inline void toROSMsgXYZ(const PointCloud<PointXYZ>* cloud, int n, const std_msgs::Header& header, sensor_msgs::PointCloud2& m) {
    static_assert(sizeof(PointXYZ) == 16, "pcl::PointXYZ is 16 bytes");
    m = sensor_msgs::PointCloud2();
    m.header = header;
    m.height = 1;
    m.width = (uint32_t)n;
    m.is_bigendian = false;
    m.is_dense = true;
    m.point_step = 16;
    m.row_step = 16u * (uint32_t)n;
    m.fields.resize(3);
    const char* names[3] = {"x", "y", "z"};
    for (int k = 0; k < 3; ++k) { m.fields[k].name = names[k]; m.fields[k].offset = 4u * (uint32_t)k; m.fields[k].datatype = sensor_msgs::PointField::FLOAT32; m.fields[k].count = 1; }
    m.data.resize((size_t)n * 16);
    if (cloud) std::memcpy(m.data.data(), cloud->points.data(), (size_t)n * 16);
    else for (int i = 0; i < n; ++i) { const PointXYZ p; std::memcpy(&m.data[(size_t)i * 16], &p, 16); }
}
inline void toROSMsgXYZ3(const float* xyz, int n, const std_msgs::Header& header, sensor_msgs::PointCloud2& m) {   // packed x y z triples
    toROSMsgXYZ(nullptr, n, header, m);
    for (int i = 0; i < n; ++i) std::memcpy(&m.data[(size_t)i * 16], xyz + 3 * (size_t)i, 12);
}
}  // namespace pclhip
