#pragma once
// object_detection/srv/ObjectDetection.srv: uint8 object_id --- bool success
#include <cstdint>
namespace object_detection { struct ObjectDetection { struct Request { uint8_t object_id = 0; }; struct Response { bool success = false; }; Request request; Response response; }; }
