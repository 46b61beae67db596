#pragma once
#include <array>
#include <memory>
#include <vector>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct CameraInfo {
    typedef std::shared_ptr<const CameraInfo> ConstPtr;
    std_msgs::Header header;
    uint32_t height = 0, width = 0;
    std::vector<double> D;
    std::array<double, 9> K{}, R{};
    std::array<double, 12> P{};
};
typedef std::shared_ptr<const CameraInfo> CameraInfoConstPtr;
}  // namespace sensor_msgs
