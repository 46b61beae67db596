#pragma once
#include <cstdint>
#include <string>
namespace ros {
struct Time { uint32_t sec = 0, nsec = 0; static Time now() { return Time(); } };
struct Duration { int32_t sec = 0, nsec = 0; Duration() {} explicit Duration(double s) : sec((int32_t)s), nsec((int32_t)((s - (int32_t)s) * 1e9)) {} };
}
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
