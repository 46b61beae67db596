#pragma once
#include <memory>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { typedef std::shared_ptr<const Pose> ConstPtr; Point position; Quaternion orientation; };
typedef std::shared_ptr<const Pose> PoseConstPtr;
}  // namespace geometry_msgs
