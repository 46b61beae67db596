#pragma once
#include <memory>
#include <vector>
#include <std_msgs/Header.h>
namespace pcl_msgs {
struct ModelCoefficients { typedef std::shared_ptr<const ModelCoefficients> ConstPtr; std_msgs::Header header; std::vector<float> values; };
typedef std::shared_ptr<const ModelCoefficients> ModelCoefficientsConstPtr;
}  // namespace pcl_msgs
