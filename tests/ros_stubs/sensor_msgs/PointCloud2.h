#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct PointField {
    enum { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
    std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0;
};
struct PointCloud2 {
    typedef std::shared_ptr<PointCloud2> Ptr;
    typedef std::shared_ptr<const PointCloud2> ConstPtr;
    std_msgs::Header header;
    uint32_t height = 0, width = 0;
    std::vector<PointField> fields;
    bool is_bigendian = false;
    uint32_t point_step = 0, row_step = 0;
    std::vector<uint8_t> data;
    bool is_dense = false;
};
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
typedef std::shared_ptr<PointCloud2> PointCloud2Ptr;
}  // namespace sensor_msgs
