#pragma once
// cuboid_detection/msg/Rectangle.msg: int64 x1, y1, x2, y2
#include <cstdint>
#include <memory>
namespace cuboid_detection { struct Rectangle { typedef std::shared_ptr<const Rectangle> ConstPtr; int64_t x1 = 0, y1 = 0, x2 = 0, y2 = 0; }; }
