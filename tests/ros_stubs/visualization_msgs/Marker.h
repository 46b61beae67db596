#pragma once
// visualization_msgs/Marker: the members object_pose_detection's publish_grasp_marker sets (opd.cpp:96-136)
#include <cstdint>
#include <string>
#include <geometry_msgs/Pose.h>
#include <ros/ros.h>
#include <std_msgs/Header.h>
namespace geometry_msgs { struct Vector3 { double x = 0, y = 0, z = 0; }; }
namespace std_msgs { struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; }; }
namespace visualization_msgs {
struct Marker {
    enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3 };
    enum { ADD = 0, MODIFY = 0, DELETE = 2 };
    std_msgs::Header header;
    std::string ns;
    int32_t id = 0, type = 0, action = 0;
    geometry_msgs::Pose pose;
    geometry_msgs::Vector3 scale;
    std_msgs::ColorRGBA color;
    ros::Duration lifetime;
    bool frame_locked = false;
};
}  // namespace visualization_msgs
