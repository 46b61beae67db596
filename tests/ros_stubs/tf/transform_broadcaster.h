#pragma once
#include <string>
#include <vector>
#include <ros/ros.h>
namespace tf {
struct Vector3 { double x, y, z; Vector3(double a = 0, double b = 0, double c = 0) : x(a), y(b), z(c) {} };
struct Quaternion { double x, y, z, w; Quaternion(double a = 0, double b = 0, double c = 0, double d = 1) : x(a), y(b), z(c), w(d) {} };
struct Transform { Quaternion q; Vector3 t; Transform() {} Transform(const Quaternion& q_, const Vector3& t_) : q(q_), t(t_) {} };
struct StampedTransform : Transform {
    ros::Time stamp; std::string frame_id, child_frame_id;
    StampedTransform(const Transform& tr, const ros::Time& s, const std::string& f, const std::string& c) : Transform(tr), stamp(s), frame_id(f), child_frame_id(c) {}
};
class TransformBroadcaster {
  public:
    void sendTransform(const StampedTransform& t) { sent().push_back(t); }
    static std::vector<StampedTransform>& sent() { static std::vector<StampedTransform> s; return s; }
};
}  // namespace tf
