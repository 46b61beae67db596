#pragma once
#include <cstdint>
#include <string>
namespace ros { struct Time { uint32_t sec = 0, nsec = 0; static Time now() { return Time(); } }; }
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
